// sim_release.h -- part of sim_device.h (included by it, in this order): extrude and release_lefs.
#pragma once

namespace modle_dev {

// =============================================================================================
// extrude + release_lefs (reference: simulation.cpp:498-521, 553-601)
// =============================================================================================
MODLE_DEV_NOINLINE void phase_extrude_and_release(Cell& c, bool burnin_completed) {
  Workspace& ws = c.ws;
  const Params& p = *c.p;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  const f64 base_p = burnin_completed ? p.p_release : p.p_release_burnin;
  const f64 affinity_soft = 1.0 / p.soft_stall_mult, affinity_hard = 1.0 / p.hard_stall_mult;
  const u32 nblk = (n + 255) / 256;
  // release_lefs draws one Bernoulli per bound LEF in LEF-id order, with a probability that
  // depends on how many of the LEF's two units are stalled by a barrier blocking their own
  // direction ("hard" stalls: 0, 1 or 2).  Every active LEF is bound at this point of the epoch, so
  // when none of the three probabilities is 0 (a zero probability consumes no draw) the draw of
  // LEF i is the raw at (stream position) + i whatever the stalls are: the three possible outcomes
  // of every LEF are evaluated FIRST, from the stream alone, and only the LEFs that are released
  // under at least one of them (a few per cent: the candidates) need their stall count.  The
  // extrusion sweep below, which passes over the ids of all units anyway, reports rank and hard
  // stall of the candidates' units (ws.r_rank / ws.f_rank, bit 31 = hard stall), found with a
  // bitmap of the candidate ids in LDS.  No per-LEF stall counters, no sweep over the LEFs, and the
  // ranks of the released LEFs -- all that select_and_bind_lefs needs in the next epoch -- come out
  // of it as well.
  const f64 prob_by_stalls[3] = {1.0 * base_p, affinity_soft * base_p, affinity_hard * base_p};
  const bool fast = prob_by_stalls[0] != 0.0 && prob_by_stalls[1] != 0.0 && prob_by_stalls[2] != 0.0;
  u32* cand = ws.tmp[2];  // candidates in id order: id | outcomes << 24 (bit s: released with s stalls)
  u32 n_cand = 0;
  if (fast) {
    const f64 thr0 = wave::uniform(prob_by_stalls[0] * TWO64), thr1 = wave::uniform(prob_by_stalls[1] * TWO64),
              thr2 = wave::uniform(prob_by_stalls[2] * TWO64);
    rank_filter_clear(c, n);
    for (u32 t = 0; t < nblk; ++t) {
      const u32 first = 256 * t;
      const u32 cnt = umin(256u, n - first);
      rng_ensure(c.g, cnt);
#pragma unroll
      for (u32 q = 0; q < 4; ++q) {
        if (first + 64 * q >= n) break;
        const u32 i = first + 64 * q + lane;
        const u64 raw = rng_peek(c.g, c.g.pos + 64 * q + lane);
        u32 code = 0;
        if (i < n) {
          const f64 x = static_cast<f64>(raw);  // bernoulli_raw with the products kept in scalar registers
          code = (x <= thr0 ? 1u : 0u) | (x <= thr1 ? 2u : 0u) | (x <= thr2 ? 4u : 0u);
        }
        const u64 m = wave::ballot(code != 0);
        if (m != 0) {
          rank_filter_add_mask(c, first + 64 * q, m);
          if (code != 0) cand[n_cand + static_cast<u32>(wave::popc64(m & lanemask_lt(lane)))] = i | (code << 24);
          n_cand += static_cast<u32>(wave::popc64(m));
        }
      }
      rng_advance(c.g, cnt);
    }
    wave::sync_lds();
  }
  // extrude in rank order, four consecutive ranks per lane (128-bit accesses; rev and fwd units of
  // the same ranks in one step: their loads are independent).  The collision words are consumed
  // here, so they are cleared on the way (the next epoch starts with clean arrays).  The loads of
  // the next block are issued before the stores of the current one: a wait for a load also waits
  // for every store issued before it.
  struct UnitRegs {
    wave::U32x4 rP, rM, rc, rI, fP, fM, fc, fI;
  };
  const auto load_units = [&](u32 t, UnitRegs& r) {
    const u32 w = 256 * t + 4 * lane;
    const u32 wq = w < n ? w : 0u;
    r.rP = wave::ld4(ws.r_pos, wq);
    r.rM = wave::ld4(ws.r_move, wq);
    r.rc = wave::ld4(ws.r_coll, wq);
    r.rI = wave::ld4(ws.r_id, wq);
    r.fP = wave::ld4(ws.f_pos, wq);
    r.fM = wave::ld4(ws.f_move, wq);
    r.fc = wave::ld4(ws.f_coll, wq);
    r.fI = wave::ld4(ws.f_id, wq);
  };
  u32 run_max_r = 0, run_max_f = 0;  // highest position after the move among the units of lower rank
  u32 n_disp_r = 0, n_disp_f = 0;
  u64* const disp_keys_r = reinterpret_cast<u64*>(ws.tmp[6]);
  u64* const disp_keys_f = reinterpret_cast<u64*>(ws.tmp[7]);
  const u32 disp_cap = umin(RANK_KEY_CAP_BIG, ws.capacity_lefs / 2);
  const auto process_block = [&](const UnitRegs& g, u32 t) {
      const u32 w = 256 * t + 4 * lane;
      {
        wave::U32x4 nr, nf;
        bool rc_any = false, fc_any = false;
#pragma unroll
        for (u32 q = 0; q < 4; ++q) {
          const bool act = w + q < n;
          const bool rb = act && g.rP.v[q] != UNBOUND, fb = act && g.fP.v[q] != UNBOUND;
          nr.v[q] = rb ? g.rP.v[q] - g.rM.v[q] : g.rP.v[q];
          nf.v[q] = fb ? g.fP.v[q] + g.fM.v[q] : g.fP.v[q];
          rc_any = rc_any || (act && g.rc.v[q] != 0);
          fc_any = fc_any || (act && g.fc.v[q] != 0);
          const bool r_hard = rb && cw_occurred_as(g.rc.v[q], EV_LEF_BAR) && (g.rc.v[q] & CW_HARD);
          const bool f_hard = fb && cw_occurred_as(g.fc.v[q], EV_LEF_BAR) && (g.fc.v[q] & CW_HARD);
          if (fast) {
            if (act && rank_filter_test(c, g.rI.v[q])) ws.r_rank[g.rI.v[q]] = (w + q) | (r_hard ? RANK_HARD : 0u);
            if (act && rank_filter_test(c, g.fI.v[q])) ws.f_rank[g.fI.v[q]] = (w + q) | (f_hard ? RANK_HARD : 0u);
          } else {
            // general form: hard stalls are counted per LEF (the sweep over the LEFs below reads them)
            if (r_hard) wave::atomic_inc_u32(&ws.stall[g.rI.v[q]]);
            if (f_hard) wave::atomic_inc_u32(&ws.stall[g.fI.v[q]]);
          }
        }
        if (w + 3 < n) {
          wave::st4(ws.r_pos, w, nr);
          wave::st4(ws.f_pos, w, nf);
          const wave::U32x4 zero = wave::zero4();
          if (rc_any) wave::st4(ws.r_coll, w, zero);
          if (fc_any) wave::st4(ws.f_coll, w, zero);
        } else {
#pragma unroll
          for (u32 q = 0; q < 4; ++q) {
            if (w + q < n) {
              ws.r_pos[w + q] = nr.v[q];
              ws.f_pos[w + q] = nf.v[q];
              if (g.rc.v[q] != 0) ws.r_coll[w + q] = 0;
              if (g.fc.v[q] != 0) ws.f_coll[w + q] = 0;
            }
          }
        }
        // units that end up below a unit of lower rank: marked and listed for the next rank update
        u32 mr[4], mf[4];
#pragma unroll
        for (u32 q = 0; q < 4; ++q) {
          const bool act = w + q < n;
          const u32 pr = (act && nr.v[q] != UNBOUND) ? nr.v[q] : 0u, pf = (act && nf.v[q] != UNBOUND) ? nf.v[q] : 0u;
          mr[q] = q == 0 ? pr : umax(mr[q - 1], pr);
          mf[q] = q == 0 ? pf : umax(mf[q - 1], pf);
        }
        const u32 sr = wave_prefix_max_u32(mr[3]), sf = wave_prefix_max_u32(mf[3]);
        const u32 sr_prev = wave::shfl_up1(sr), sf_prev = wave::shfl_up1(sf);
        const u32 excl_r = umax(run_max_r, lane > 0 ? sr_prev : 0u), excl_f = umax(run_max_f, lane > 0 ? sf_prev : 0u);
        run_max_r = umax(run_max_r, wave::bcast(sr, 63));
        run_max_f = umax(run_max_f, wave::bcast(sf, 63));
        bool dr[4], df[4];
        bool any_d = false;
#pragma unroll
        for (u32 q = 0; q < 4; ++q) {
          const bool act = w + q < n;
          dr[q] = act && nr.v[q] != UNBOUND && nr.v[q] < (q == 0 ? excl_r : umax(excl_r, mr[q - 1]));
          df[q] = act && nf.v[q] != UNBOUND && nf.v[q] < (q == 0 ? excl_f : umax(excl_f, mf[q - 1]));
          any_d = any_d || dr[q] || df[q];
        }
        if (wave::any(any_d)) {
#pragma unroll
          for (u32 q = 0; q < 4; ++q) {
            const u64 mr_ = wave::ballot(dr[q]), mf_ = wave::ballot(df[q]);
            if (dr[q]) {
              const u32 e = n_disp_r + static_cast<u32>(wave::popc64(mr_ & lanemask_lt(lane)));
              if (e < disp_cap) disp_keys_r[e] = (static_cast<u64>(nr.v[q]) << 32) | (w + q);
              ws.r_move[w + q] = DISP_MARK;
            }
            if (df[q]) {
              const u32 e = n_disp_f + static_cast<u32>(wave::popc64(mf_ & lanemask_lt(lane)));
              if (e < disp_cap) disp_keys_f[e] = (static_cast<u64>(nf.v[q]) << 32) | (w + q);
              ws.f_move[w + q] = DISP_MARK;
            }
            n_disp_r += static_cast<u32>(wave::popc64(mr_));
            n_disp_f += static_cast<u32>(wave::popc64(mf_));
          }
        }
      }
  };
  {
    // two blocks of loads in flight: the sweep is bound by the latency of its loads, not by what it
    // does with them
    UnitRegs ra, rb;
    load_units(0, ra);
    if (1 < nblk) load_units(1, rb);
    for (u32 t = 0; t < nblk; t += 2) {
      {
        const UnitRegs g = ra;
        if (t + 2 < nblk) load_units(t + 2, ra);
        process_block(g, t);
      }
      if (t + 1 < nblk) {
        const UnitRegs g = rb;
        if (t + 3 < nblk) load_units(t + 3, rb);
        process_block(g, t + 1);
      }
    }
  }
  c.n_disp[0] = n_disp_r;
  c.n_disp[1] = n_disp_f;
  c.disp_valid = n_disp_r <= disp_cap && n_disp_f <= disp_cap;
  wave::sync_mem();
  u32* list = reinterpret_cast<u32*>(c.lds.sort_lds);
  u32 n_rel = 0;
  if (fast) {
    // the candidates whose outcome for their number of stalls is "released", in id order: listed
    // in LDS for the next epoch's select_and_bind_lefs, units and binding epoch marked
    wave::lockstep();
    for (u32 base = 0; base < n_cand; base += 64) {
      const u32 e = base + lane;
      const bool act = e < n_cand;
      const u32 cw = wave::ld_sel(cand, e, act, 0u);
      const u32 id = cw & 0x00FFFFFFu;
      const u32 rw = wave::ld_sel(ws.r_rank, id, act, 0u), fw = wave::ld_sel(ws.f_rank, id, act, 0u);
      const u32 stalls = (rw >> 31) + (fw >> 31);
      const bool rel = act && (((cw >> 24) >> stalls) & 1u) != 0;
      const u64 rm = wave::ballot(rel);
      const u32 kr = rw & ~RANK_HARD, kf = fw & ~RANK_HARD;
      if (act) {
        // (without the stall flag: the entries stay valid ranks -- the next bind reads those of the
        // released LEFs, and a complete inverse permutation stays complete)
        ws.r_rank[id] = kr;
        ws.f_rank[id] = kf;
      }
      if (rel) {
        const u32 j = n_rel + static_cast<u32>(wave::popc64(rm & lanemask_lt(lane)));
        if (j < REL_CAP) list[j] = id;
        ws.epoch[id] = UNBOUND;
        ws.r_pos[kr] = UNBOUND;
        ws.f_pos[kf] = UNBOUND;
      }
      n_rel += static_cast<u32>(wave::popc64(rm));
    }
    wave::sync_lds();
    c.rel_valid = n_rel <= REL_CAP;
    c.n_rel = c.rel_valid ? n_rel : 0;
#ifdef MODLE_EMU_TRACE_RANK  // (emulator only: which regime a test exercises)
    if (lane == 0 && !c.rel_valid) fprintf(stderr, "release: %u LEFs released, the list holds %u: the next bind sweeps\n", n_rel, REL_CAP);
#endif
    wave::sync_mem();
    return;
  }
  // General form (a release probability of zero): draws in LEF-id order, four consecutive ids per
  // lane: the draw of a LEF is the raw at (stream position) + (bound LEFs with a non-zero
  // probability before it).  The released LEFs are listed in LDS; their units are marked afterwards
  // from the list, and the next epoch's select_and_bind_lefs binds from the same list.
  ensure_inverse_both(c);
  struct LefRegs {
    wave::U32x4 E, H;
  };
  const auto load_lefs = [&](u32 t, LefRegs& r) {
    const u32 w = 256 * t + 4 * lane;
    const u32 wq = w < n ? w : 0u;
    r.E = wave::ld4(ws.epoch, wq);
    r.H = wave::ld4(ws.stall, wq);
  };
  wave::lockstep();
  LefRegs lcur;
  load_lefs(0, lcur);
  for (u32 t = 0; t < nblk; ++t) {
    const LefRegs g = lcur;
    if (t + 1 < nblk) load_lefs(t + 1, lcur);
    const u32 w = 256 * t + 4 * lane;
    f64 prob[4];
    bool draws[4];
    u32 before[4];  // draws of this lane before LEF q
    u32 lane_draws = 0;
    bool hard_any = false;
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      const bool act = w + q < n;
      const u32 hard = g.H.v[q];
      hard_any = hard_any || (act && hard != 0);
      const f64 affinity = hard == 0 ? 1.0 : (hard == 1 ? affinity_soft : affinity_hard);
      prob[q] = act ? affinity * base_p : 0.0;
      draws[q] = act && g.E.v[q] != UNBOUND && prob[q] != 0.0;
      before[q] = lane_draws;
      lane_draws += draws[q] ? 1u : 0u;
    }
    if (hard_any) {
      if (w + 3 < n) {
        const wave::U32x4 zero = wave::zero4();
        wave::st4(ws.stall, w, zero);
      } else {
#pragma unroll
        for (u32 q = 0; q < 4; ++q) {
          if (w + q < n && g.H.v[q] != 0) ws.stall[w + q] = 0;
        }
      }
    }
    const u32 ps = wave_prefix_sum_u32(lane_draws);
    const u32 cnt = wave::bcast(ps, 63);
    const u32 lane_first = ps - lane_draws;
    rng_ensure(c.g, cnt);
    bool rel[4];
    u32 rel_before[4];
    u32 lane_rel = 0;
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      rel[q] = draws[q] && bernoulli_raw(rng_peek(c.g, c.g.pos + lane_first + before[q]), prob[q]);
      rel_before[q] = lane_rel;
      lane_rel += rel[q] ? 1u : 0u;
    }
    rng_advance(c.g, cnt);
    if (wave::any(lane_rel != 0)) {
      const u32 rs = wave_prefix_sum_u32(lane_rel);
      const u32 lane_slot = n_rel + rs - lane_rel;
#pragma unroll
      for (u32 q = 0; q < 4; ++q) {
        if (rel[q]) {
          const u32 i = w + q;
          const u32 j = lane_slot + rel_before[q];
          if (j < REL_CAP) {
            list[j] = i;  // (its epoch and units are marked from the list, after the sweep: no
                          // store here that the wait for the next block's loads would include)
          } else {
            // more releases than the list holds (the next bind then sweeps the LEFs instead)
            ws.epoch[i] = UNBOUND;
            ws.r_pos[ws.r_rank[i]] = UNBOUND;
            ws.f_pos[ws.f_rank[i]] = UNBOUND;
          }
        }
      }
      n_rel += wave::bcast(rs, 63);
    }
  }
  wave::sync_lds();
  c.rel_valid = n_rel <= REL_CAP;
  c.n_rel = c.rel_valid ? n_rel : 0;
  const u32 n_listed = umin(n_rel, REL_CAP);
  for (u32 base = 0; base < n_listed; base += 64) {
    const u32 e = base + lane;
    if (e < n_listed) {
      const u32 id = list[e];
      ws.epoch[id] = UNBOUND;
      ws.r_pos[ws.r_rank[id]] = UNBOUND;
      ws.f_pos[ws.f_rank[id]] = UNBOUND;
    }
  }
  wave::sync_mem();
}

}  // namespace modle_dev

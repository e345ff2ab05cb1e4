// sim_collisions.h -- part of sim_device.h (included by it, in this order): process_collisions: boundaries, LEF-BAR, primary and secondary LEF-LEF collisions, fix_secondary.
#pragma once

namespace modle_dev {

// =============================================================================================
// process_collisions (reference: simulation.cpp:763-793 and simulation_detect_collisions.cpp)
// =============================================================================================
struct BoundaryCounts {
  u32 n5, n3;
};

// detect_units_at_interval_boundaries (reference: simulation_detect_collisions.cpp:25-120)
MODLE_DEV_NOINLINE BoundaryCounts detect_boundaries(Cell& c) {
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  const u32 start = c.iv->start, last = c.iv->end - 1;
  const u32 first_fwd_pos = ws.f_pos[0];
  // position of the last bound unit in rev rank order
  u32 last_rev_pos = 0;
  for (u32 top = n; top > 0;) {
    const u32 cnt = umin(64u, top);
    const bool act = lane < cnt;
    const u32 P = wave::ld_sel(ws.r_pos, top - 1 - lane, act, UNBOUND);  // descending ranks
    const u64 m = wave::ballot(act && P != UNBOUND);
    if (m != 0) {
      last_rev_pos = wave::bcast(P, static_cast<u32>(wave::ctz64(m)));
      break;
    }
    top -= cnt;
  }
  BoundaryCounts out{0, 0};
  const u32 mark5 = cw_make(5, EV_COLLISION | EV_CHROM_BOUNDARY);
  const u32 mark3 = cw_make(3, EV_COLLISION | EV_CHROM_BOUNDARY);
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    const bool act = k < n;
    const u32 P = wave::ld_sel(ws.r_pos, k, act, 0);
    const u32 M = wave::ld_sel(ws.r_move, k, act, 0);
    const bool at = act && P == start;
    const bool brk_b = act && !at && P > first_fwd_pos;
    const bool brk_c = act && !at && !brk_b && P - M == start;
    const u64 stop = wave::ballot(brk_b || brk_c);
    const u32 s = stop != 0 ? static_cast<u32>(wave::ctz64(stop)) : 64u;
    const bool mark = act && ((lane < s && at) || (lane == s && brk_c));
    if (mark) ws.r_coll[k] = mark5;
    out.n5 += static_cast<u32>(wave::popc64(wave::ballot(mark)));
    if (stop != 0) break;
  }
  // fwd units: ranks n-1 down to 1 (rank 0 is never visited, simulation_detect_collisions.cpp:91)
  for (u32 top = n; top > 1;) {
    const u32 cnt = umin(64u, top - 1);
    const u32 k = top - 1 - lane;
    const bool act = lane < cnt;
    const u32 P = wave::ld_sel(ws.f_pos, k, act, 0);
    const u32 M = wave::ld_sel(ws.f_move, k, act, 0);
    const bool bnd = act && P != UNBOUND;
    const bool unb = act && !bnd;
    const bool at = bnd && P == last;
    const bool brk_b = bnd && !at && P < last_rev_pos;
    const bool brk_c = bnd && !at && !brk_b && P + M == last;
    const u64 stop = wave::ballot(brk_b || brk_c);
    const u32 s = stop != 0 ? static_cast<u32>(wave::ctz64(stop)) : 64u;
    const bool mark = (lane < s && at) || (lane == s && brk_c);
    if (mark) ws.f_coll[k] = mark3;
    out.n3 += static_cast<u32>(wave::popc64(wave::ballot(mark || (lane < s && unb))));
    if (stop != 0) break;
    top -= cnt;
  }
  wave::sync_mem();
  return out;
}

// detect_lef_bar_collisions (reference: simulation_detect_collisions.cpp:123-247), evaluated
// per extrusion unit: barrier b is tested against the first rev unit downstream of it (first fwd
// unit upstream), so the barriers that can stall the unit of rank j are those between the unit
// of rank j-1 and itself that lie within its move.  Bernoulli trials (pblock not in {0,1}) are
// numbered in the reference's order: barriers ascending for rev units, descending for fwd units.
//
// Only active barriers whose blocking probability is not zero can stall a unit: they are compacted
// once per epoch, in position order, into one list per direction (sim_barriers.h: position,
// index | HITBAR_HARD), and a window of BAR_WIN list entries is staged in LDS (the sort buffer, idle
// during the collision passes) and moved along the list as the ranks advance.  Unit windows are
// disjoint and ordered like the ranks, so every block of units continues the search where the
// previous one stopped; a block whose windows do not fit the staged entries searches the list in
// device memory.
//   * Blocking probabilities in {0, 1} (the reference default: major 1, minor 0): no trial is drawn,
//     every entry stalls, and of the entries in a unit's window the reference keeps the one it visits
//     last -- the highest for a rev unit, the lowest for a fwd unit: one search and one test per unit.
//   * Otherwise (BASELINE configs[4]: minor 0.3) the same sweep only LISTS the units whose window
//     holds an entry -- a few per cent of them -- and a second pass takes 64 listed units at a time:
//     window ends, trials numbered by a prefix sum in list order = draw order, outcomes, the entry
//     visited last among the hits.  (Round 4.  Before, every configuration with a fractional
//     probability searched the complete barrier set per unit, one rank per lane, and counted and
//     drew in two per-lane loops over every barrier of the window: 30 % of a configs[4] launch.)
constexpr u32 BAR_WIN = SORT_LDS_CAP;  // SORT_LDS_CAP u64 keys = 2 * BAR_WIN words

// Position of the barrier that stalls the unit of rank k (valid where the collision word says
// LEF-BAR), written by detect_lef_bar for the passes that correct moves.  Lives in ranking
// scratch, which is idle during the collision passes.
template <bool FWD>
MODLE_DEV u32* stalling_barrier_positions(const Workspace& ws) {
  return FWD ? ws.tmp[4] : ws.tmp[3];
}

constexpr u32 HITBAR_NEAR = 127;

// Copies `cnt` (<= BAR_WIN) list entries into the LDS window: all loads in flight, then the LDS
// writes.  A real call: it runs a few times per pass and its registers stay out of the pass's
// allocation.
MODLE_DEV_CALL void stage_stalling_window_call(MODLE_LDS u32* cp, MODLE_LDS u32* ci,
                                               const u32* hpos, const u32* hidx, u32 cnt) {
  const u32 lane = wave::lane();
  const u32* gp = wave::as_global(hpos);
  const u32* gi = wave::as_global(hidx);
  u32 Hp[BAR_WIN / 64], Hi[BAR_WIN / 64];
#pragma unroll
  for (u32 t = 0; t < BAR_WIN / 64; ++t) {
    const u32 e = lane + 64 * t;
    Hp[t] = e < cnt ? gp[e] : 0;
    Hi[t] = e < cnt ? gi[e] : 0;
  }
#pragma unroll
  for (u32 t = 0; t < BAR_WIN / 64; ++t) {
    const u32 e = lane + 64 * t;
    if (e < cnt) {
      cp[e] = Hp[t];
      ci[e] = Hi[t];
    }
  }
}  // entries next to the anchor that the fixed-step search covers

constexpr u32 BAR_NEED = 128;
// Barriers [s0, s1) are staged.  STAGED_ONLY accessors assume the index is inside the staged
// range (the caller has checked that the whole batch stays inside); the general ones read
// everything else from device memory.
struct BarView {
  const Interval* iv;
  const u8* active;
  const u32* st_pos;
  const u32* st_flag;
  u32 s0, s1;
  template <bool STAGED_ONLY>
  MODLE_DEV_MEMBER u32 pos(u32 b) const {
    if (STAGED_ONLY) return st_pos[b - s0];
    return (b >= s0 && b < s1) ? st_pos[b - s0] : iv->bar_pos[b];
  }
  // bit 0: active, bits 1..2: blocking direction
  template <bool STAGED_ONLY>
  MODLE_DEV_MEMBER u32 flag(u32 b) const {
    if (STAGED_ONLY) return st_flag[b - s0];
    return (b >= s0 && b < s1) ? st_flag[b - s0]
                               : (static_cast<u32>(active[b] != 0) | (static_cast<u32>(iv->bar_dir[b]) << 1));
  }
};

// first barrier index in [lo, hi) whose position is >= key (hi when there is none)
template <bool STAGED_ONLY>
MODLE_DEV u32 bar_view_lower_bound(const BarView& v, u32 lo, u32 hi, u64 key) {
  while (lo < hi) {
    const u32 mid = (lo + hi) >> 1;
    if (v.pos<STAGED_ONLY>(mid) < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Barrier index window [b_lo, b_hi) of one unit: lo_key <= position < hi_key.  With STAGED_ONLY
// the search stays inside the staged range and reports `edge` when the answer touches an edge
// beyond which more barriers exist (the batch is then redone with the general accessors).
template <bool FWD, bool STAGED_ONLY>
MODLE_DEV void lef_bar_window(const BarView& v, u32 nb, u32 anchor, u64 lo_key, u64 hi_key,
                              u32& b_lo, u32& b_hi, bool& edge) {
  edge = false;
  if (!FWD) {
    if (STAGED_ONLY) {
      // the answer is almost always within a few dozen barriers of the batch's anchor
      const u32 near = umin(anchor + BAR_NEED, v.s1);
      b_lo = bar_view_lower_bound<true>(v, anchor, near, lo_key);
      if (b_lo == near && near < v.s1) b_lo = bar_view_lower_bound<true>(v, near, v.s1, lo_key);
    } else {
      b_lo = bar_view_lower_bound<false>(v, v.s0, v.s1, lo_key);
    }
    if (b_lo == v.s1 && v.s1 < nb) {
      if (STAGED_ONLY) {
        edge = true;
        b_hi = b_lo;
        return;
      }
      b_lo = bar_view_lower_bound<false>(v, v.s1, nb, lo_key);
    }
    b_hi = b_lo;
    const u32 lim = STAGED_ONLY ? v.s1 : nb;
    while (b_hi < lim && v.pos<STAGED_ONLY>(b_hi) < hi_key) ++b_hi;
    if (STAGED_ONLY && b_hi == v.s1 && v.s1 < nb) edge = true;
  } else {
    if (STAGED_ONLY) {
      const u32 near = anchor > v.s0 + BAR_NEED ? anchor - BAR_NEED : v.s0;
      b_hi = bar_view_lower_bound<true>(v, near, anchor, hi_key);
      if (b_hi == near && near > v.s0) b_hi = bar_view_lower_bound<true>(v, v.s0, near, hi_key);
    } else {
      b_hi = bar_view_lower_bound<false>(v, v.s0, v.s1, hi_key);
    }
    if (b_hi == v.s0 && v.s0 > 0) {
      if (STAGED_ONLY) {
        edge = true;
        b_lo = b_hi;
        return;
      }
      b_hi = bar_view_lower_bound<false>(v, 0, v.s0, hi_key);
    }
    b_lo = b_hi;
    const u32 lim = STAGED_ONLY ? v.s0 : 0;
    while (b_lo > lim && v.pos<STAGED_ONLY>(b_lo - 1) >= lo_key) --b_lo;
    if (STAGED_ONLY && b_lo == v.s0 && v.s0 > 0) edge = true;
  }
}

// the barrier that stalls one unit, among barriers [b_lo, b_hi) of a view (no trials on this path)
template <bool FWD, bool STAGED_ONLY>
MODLE_DEV u32 lef_bar_pick(const BarView& v, const Params& p, const Rng& g, u32 b_lo, u32 b_hi,
                           u32 trial_off, bool& hard, u32& bpos) {
  const u32 major_dir = FWD ? DIR_FWD : DIR_REV;
  u32 winner = 0xFFFFFFFFu;
  u32 t = 0;
  for (u32 q = b_lo; q < b_hi; ++q) {
    const u32 b = FWD ? (b_hi - 1 - (q - b_lo)) : q;  // reference visiting order
    const u32 fl = v.flag<STAGED_ONLY>(b);
    if (!(fl & 1u)) continue;
    const f64 pb = (fl >> 1) == major_dir ? p.pblock_major : p.pblock_minor;
    bool hit;
    if (pb == 1.0) {
      hit = true;
    } else if (pb == 0.0) {
      hit = false;
    } else {
      hit = bernoulli_raw(rng_peek(g, g.pos + trial_off + t), pb);
      ++t;
    }
    if (hit) {  // later visits overwrite earlier ones
      winner = b;
      hard = (fl >> 1) == major_dir;
    }
  }
  if (winner != 0xFFFFFFFFu) bpos = v.pos<STAGED_ONLY>(winner);
  return winner;
}

// ---- blocking probabilities in {0, 1}: the sweep of rounds 2-3, unchanged (the headline path) ----
// (its fall-back for a block that spans more list entries than the window holds searches the
// complete barrier set through a BarView: a few units spread over a chromosome, early burn-in)
template <bool FWD>
MODLE_DEV_NOINLINE void detect_lef_bar_det(Cell& c, BoundaryCounts bc) {
  Workspace& ws = c.ws;
  const Interval& iv = *c.iv;
  const Params& p = *c.p;
  const u32 n = wave::uniform(c.n_active);
  const u32 nb = wave::uniform(iv.n_barriers);
  const bool major_hits = p.pblock_major == 1.0, minor_hits = p.pblock_minor == 1.0;
  if (!major_hits && !minor_hits) return;  // no barrier ever stalls a unit
  const u32 major_dir = FWD ? DIR_FWD : DIR_REV;
  const u32 lane = wave::lane();
  const u32* pos = FWD ? ws.f_pos : ws.r_pos;
  const move_t* moves = FWD ? ws.f_move : ws.r_move;
  u32* coll = FWD ? ws.f_coll : ws.r_coll;
  u32* barpos = stalling_barrier_positions<FWD>(ws);
  // positions of the compacted barriers and their indices (| HITBAR_HARD): at most BAR_FILL entries,
  // with a sentinel next to them so that the fixed-step searches need no range test -- rev: a word
  // of all ones behind the last entry (cp[cnt]; a step that overshoots reads it through one
  // `v_min` on the index); fwd: a zero in front of the first entry (the window starts one word into
  // the buffer; a step that undershoots reads cp[-1] through one `v_max`)
  constexpr u32 BAR_FILL = BAR_WIN - 1;
  u32* cp = reinterpret_cast<u32*>(c.lds.sort_lds) + (FWD ? 1 : 0);
  u32* ci = cp + BAR_WIN;
  const u32 j_rev0 = bc.n5 == 0 ? 0 : bc.n5 - 1;
  const u32 j_fwd0 = bc.n3 == 0 ? n - 1 : n - bc.n3;
  u32 carry_pos = 0;
  const u32 nh = wave::uniform(c.n_hit[FWD ? 1 : 0]);
  if (nh == 0 || n == 0) return;  // no barrier stalls a unit of this direction in this epoch
  const u32* hpos = ws.hit_pos[FWD ? 1 : 0];
  const u32* hidx = ws.hit_idx[FWD ? 1 : 0];
  constexpr u32 c0 = 0;
  u32 g0 = 0, g1 = 0, cnt = 0;      // the window holds list entries [g0, g1): cp[0 .. cnt)
  bool staged = false;
  u32 lo_cover = 1, hi_cover = 0;   // nothing staged yet (0xFFFFFFFF: no bound)
  u32 anchor = 0;                   // rev: entries below it lie before the batch; fwd: entries at
                                    // or above it lie beyond the batch (relative to c0)
  // Four consecutive ranks per lane, blocks of 256 ranks on 256-rank boundaries (128-bit loads);
  // rev: ranks ascending from j_rev0, fwd: ranks descending from j_fwd0 (lane 0 holds the highest
  // ranks of a block and walks its four units downwards).  Ranks outside the sweep are masked.
  const u32 b_first = (FWD ? j_fwd0 : j_rev0) / 256;
  const u32 nblk = FWD ? b_first + 1 : (n + 255) / 256 - b_first;
  const auto word0 = [&](u32 t) { return (FWD ? b_first - t : b_first + t) * 256 + 4 * (FWD ? 63 - lane : lane); };
  struct Blk {
    wave::U32x4 P, M;
  };
  const auto load_blk = [&](u32 t, Blk& r) {
    const u32 w = word0(t);
    const u32 wq = w < n ? w : 0u;
    r.P = wave::ld4(pos, wq);
    r.M = wave::ld4(moves, wq);
  };
  Blk cur;
  load_blk(0, cur);
  for (u32 t = 0; t < nblk; ++t) {
    const Blk g = cur;
    if (t + 1 < nblk) load_blk(t + 1, cur);
    const u32 w = word0(t);
    u32 k[4], P[4], lo_key[4], hi_key[4];
    bool bnd[4];
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {  // j: position in sweep order inside the lane
      const u32 q = FWD ? 3 - j : j;
      k[j] = w + q;
      const bool act = FWD ? k[j] <= j_fwd0 : (k[j] >= j_rev0 && k[j] < n);
      P[j] = act ? g.P.v[q] : 0u;
      bnd[j] = act && P[j] != UNBOUND;
    }
    const u32 nbr_in = wave::shfl_up1(P[3]);
    const u32 nbr0 = lane > 0 ? nbr_in : carry_pos;
    carry_pos = wave::bcast(P[3], 63);
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      const u32 q = FWD ? 3 - j : j;
      const u32 M = g.M.v[q];
      const bool first = k[j] == (FWD ? j_fwd0 : j_rev0);
      const u32 nbr = j == 0 ? nbr0 : P[j - 1];
      // see detect_lef_bar; 32-bit keys: positions lie below 2^32 - 2 (the host rejects longer
      // intervals), and a reach beyond that is as good as 2^32 - 2
      // (units that take no part: keys no window entry compares with -- nothing lies below 0,
      // nothing at or above 2^32 - 1)
      lo_key[j] = FWD ? 0xFFFFFFFFu : 0u;
      hi_key[j] = 0;
      if (bnd[j]) {
        if (!FWD) {
          const u32 reach = P[j] - M;
          lo_key[j] = first ? reach : umax(reach, nbr);
          hi_key[j] = P[j];
        } else {
          const u32 sum = P[j] + M;
          const u32 reach = (sum < P[j] || sum > 0xFFFFFFFEu) ? 0xFFFFFFFEu : sum;
          lo_key[j] = P[j] + 1;
          hi_key[j] = (first ? reach : umin(reach, nbr)) + 1;
        }
      }
    }
    const u64 bm = wave::ballot(bnd[0] || bnd[1] || bnd[2] || bnd[3]);
    if (bm == 0) continue;
    const u32 l_first = static_cast<u32>(wave::ctz64(bm));
    const u32 l_last = static_cast<u32>(63 - wave::clz64(bm));
    // keys of the lane's first / last bound unit in sweep order
    const u32 jf = bnd[0] ? 0u : bnd[1] ? 1u : bnd[2] ? 2u : 3u;
    const u32 jl = bnd[3] ? 3u : bnd[2] ? 2u : bnd[1] ? 1u : 0u;
    const u32 lo_f = jf == 0 ? lo_key[0] : jf == 1 ? lo_key[1] : jf == 2 ? lo_key[2] : lo_key[3];
    const u32 hi_f = jf == 0 ? hi_key[0] : jf == 1 ? hi_key[1] : jf == 2 ? hi_key[2] : hi_key[3];
    const u32 lo_l = jl == 3 ? lo_key[3] : jl == 2 ? lo_key[2] : jl == 1 ? lo_key[1] : lo_key[0];
    const u32 hi_l = jl == 3 ? hi_key[3] : jl == 2 ? hi_key[2] : jl == 1 ? hi_key[1] : hi_key[0];
    // keys the block spans (sweep order holds ascending positions for rev, descending for fwd)
    const u32 need_lo = FWD ? wave::bcast(lo_l, l_last) : wave::bcast(lo_f, l_first);
    const u32 need_hi = FWD ? wave::bcast(hi_f, l_first) : wave::bcast(hi_l, l_last);
#ifdef MODLE_SUBTIMER_LEFBAR
    const u64 t_stage = wave::clock();
#endif
    if (need_lo < lo_cover || need_hi > hi_cover) {
      // Move the window along the list to where this block starts (one coalesced load of
      // positions and indices).  Entries the window has already passed are dropped by counting;
      // when the block lies beyond the whole window, the window keeps moving.
      u32 moved = 0;
      for (;;) {
        if (staged) {
          // window entries before the block (rev: below need_lo; fwd: below need_hi)
          const u32 key = FWD ? need_hi : need_lo;
          u32 below = 0;
#pragma unroll
          for (u32 e0 = 0; e0 < BAR_WIN / 64; ++e0) {
            const u32 e = lane + 64 * e0;
            const u32 ce = cp[e];  // (e < BAR_WIN: inside the window whatever cnt is)
            below += static_cast<u32>(wave::popc64(wave::ballot((e < cnt) & (ce < key))));
          }
          if (!FWD) {
            g0 += below;
          } else {
            g1 = g0 + below;
          }
        } else {
          if (FWD) g1 = nh; else g0 = 0;
        }
        if (!FWD) {
          g1 = umin(g0 + BAR_FILL, nh);
        } else {
          g0 = g1 > BAR_FILL ? g1 - BAR_FILL : 0;
        }
        cnt = g1 - g0;
        wave::lockstep();
        {
          // what the window does not hold: everything before it lies below lo_cover,
          // everything after it at or above hi_cover
          const u32 edge_lo = g0 > 0 ? hpos[g0 - 1] : 0;
          const u32 edge_hi = g1 < nh ? hpos[g1] : 0;
          stage_stalling_window_call((MODLE_LDS u32*)cp, (MODLE_LDS u32*)ci, hpos + g0, hidx + g0, cnt);
          if (lane == 0) {
            if (FWD) cp[-1] = 0u; else cp[cnt] = 0xFFFFFFFFu;  // (the sentinel of the searches)
          }
          lo_cover = g0 > 0 ? wave::uniform(edge_lo) + 1 : 0;
          hi_cover = g1 < nh ? wave::uniform(edge_hi) : 0xFFFFFFFFu;
        }
        wave::sync_lds();
        staged = true;
        anchor = FWD ? cnt : 0;
        // done unless the block starts beyond this window and the list goes on
        const bool beyond = FWD ? (need_hi <= lo_cover && g0 > 0) : (need_lo >= hi_cover && g1 < nh);
        if (!beyond || ++moved > 64) break;  // (a block that is still not covered is looked up in device memory)
      }
    }
#ifdef MODLE_SUBTIMER_LEFBAR
    c.ph[14] += wave::clock() - t_stage;
    const u64 t_search = wave::clock();
#endif
    u32 winner[4], bpos[4];
    bool hard[4];
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      winner[j] = 0xFFFFFFFFu;
      bpos[j] = 0;
      hard[j] = false;
    }
    if (need_lo >= lo_cover && need_hi <= hi_cover) {
      // four searches side by side, all from the anchor the previous block left
      u32 q[4];
#pragma unroll
      for (u32 j = 0; j < 4; ++j) q[j] = anchor;
      if (!FWD) {
        // q = number of entries before the unit: the last of them is the candidate
#pragma unroll
        for (u32 sft = 64; sft >= 1; sft >>= 1) {
          // (the four reads of a round are issued together: left alone the compiler waits for each)
          u32 jx[4], kv[4];
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            jx[j] = q[j] + sft;
            kv[j] = (cp - 1)[umin(jx[j], cnt + 1)];  // (beyond the window: the sentinel)
          }
          wave::sched_fence();
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            if (kv[j] < hi_key[j]) q[j] = jx[j];
          }
          wave::sched_fence();
        }
        bool far = false;  // the fixed steps ran out: finish with a binary search (rare)
#pragma unroll
        for (u32 j = 0; j < 4; ++j) far = far || (bnd[j] && q[j] == anchor + HITBAR_NEAR && q[j] < cnt);
        if (wave::any(far)) {
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            if (bnd[j] && q[j] == anchor + HITBAR_NEAR && q[j] < cnt) {
              u32 hi = cnt;
              u32 l = q[j];
              while (l < hi) {
                const u32 mid = (l + hi) >> 1;
                if (cp[c0 + mid] < hi_key[j]) l = mid + 1; else hi = mid;
              }
              q[j] = l;
            }
          }
        }
        // the candidates of the four units: position and index read together, then tested
        u32 bp[4], wd[4];
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          const u32 e = c0 + (q[j] > 0 ? q[j] - 1 : 0);
          bp[j] = cp[e];
          wd[j] = ci[e];
        }
        wave::sched_fence();
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          if (bnd[j] & (q[j] > 0) & (bp[j] >= lo_key[j])) {
            winner[j] = wd[j] & ~HITBAR_HARD;
            hard[j] = (wd[j] & HITBAR_HARD) != 0;
            bpos[j] = bp[j];
          }
        }
      } else {
        // q = number of entries at or before the unit: entry q is the candidate
#pragma unroll
        for (u32 sft = 64; sft >= 1; sft >>= 1) {
          u32 kv[4];
          i32 tq[4];
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            tq[j] = static_cast<i32>(q[j]) - static_cast<i32>(sft);
            kv[j] = cp[tq[j] > -1 ? tq[j] : -1];  // (before the window: the sentinel)
          }
          wave::sched_fence();
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            if (kv[j] >= lo_key[j]) q[j] = static_cast<u32>(tq[j]);
          }
          wave::sched_fence();
        }
        bool far = false;  // the fixed steps ran out: finish with a binary search (rare)
#pragma unroll
        for (u32 j = 0; j < 4; ++j) far = far || (bnd[j] && q[j] + HITBAR_NEAR == anchor && q[j] > 0);
        if (wave::any(far)) {
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            if (bnd[j] && q[j] + HITBAR_NEAR == anchor && q[j] > 0) {
              u32 lo = 0;
              u32 h = q[j];
              while (lo < h) {
                const u32 mid = (lo + h) >> 1;
                if (cp[c0 + mid] < lo_key[j]) lo = mid + 1; else h = mid;
              }
              q[j] = h;
            }
          }
        }
        // the candidates of the four units: position and index read together, then tested
        u32 bp[4], wd[4];
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          const u32 e = c0 + (q[j] < cnt ? q[j] : 0);
          bp[j] = cp[e];
          wd[j] = ci[e];
        }
        wave::sched_fence();
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          if (bnd[j] & (q[j] < cnt) & (bp[j] < hi_key[j])) {
            winner[j] = wd[j] & ~HITBAR_HARD;
            hard[j] = (wd[j] & HITBAR_HARD) != 0;
            bpos[j] = bp[j];
          }
        }
      }
      const u32 q_last = jl == 3 ? q[3] : jl == 2 ? q[2] : jl == 1 ? q[1] : q[0];
      anchor = wave::bcast(q_last, l_last);
    } else {
      // the block spans more stalling barriers than the window holds (few, far apart units):
      // per-unit searches in device memory
      BarView v;
      v.iv = &iv;
      v.active = ws.bar_active;
      v.st_pos = cp;
      v.st_flag = ci;
      v.s0 = FWD ? nb : 0;  // empty staged range at the end the search starts from
      v.s1 = v.s0;
#pragma unroll
      for (u32 j = 0; j < 4; ++j) {
        u32 b_lo = 0, b_hi = 0;
        bool edge = false;
        if (bnd[j]) lef_bar_window<FWD, false>(v, nb, 0, lo_key[j], hi_key[j], b_lo, b_hi, edge);
        winner[j] = lef_bar_pick<FWD, false>(v, p, c.g, b_lo, b_hi, 0, hard[j], bpos[j]);
        // (the loads of this rare path end here: see detect_primary)
        wave::pin(winner[j]);
        wave::pin(bpos[j]);
        u32 hd = hard[j] ? 1u : 0u;
        wave::pin(hd);
        hard[j] = hd != 0;
      }
      // the staged entries stay valid, but the next block must not trust the anchor
      lo_cover = 1;
      hi_cover = 0;
    }
#ifdef MODLE_SUBTIMER_LEFBAR
    c.ph[15] += wave::clock() - t_search;
#endif
    if (wave::any((winner[0] & winner[1] & winner[2] & winner[3]) != 0xFFFFFFFFu)) {
#pragma unroll
      for (u32 j = 0; j < 4; ++j) {
        if (winner[j] != 0xFFFFFFFFu) {
          coll[k[j]] = cw_make(winner[j], EV_COLLISION | EV_LEF_BAR) | (hard[j] ? CW_HARD : 0u);
          barpos[k[j]] = bpos[j];
        }
      }
    }
  }
  wave::sync_mem();
}


// ---- fractional blocking probabilities ----
// One sweep over the units of a direction (rank order), four consecutive ranks per lane: finds for
// every unit the list entry its window ends at.  TRIALS = false: no probability needs a Bernoulli
// trial, the entry at the end of a non-empty window is the one the reference keeps, the collision
// word is written on the spot.  TRIALS = true: the units with a non-empty window are LISTED (rank,
// list entry at the near end of the window, key of its far end) in sweep order = the reference's
// draw order, for resolve_listed_units below; returns their number.
template <bool FWD, bool TRIALS>
MODLE_DEV_NOINLINE u32 detect_lef_bar_sweep(Cell& c, BoundaryCounts bc) {
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  const u32* pos = FWD ? ws.f_pos : ws.r_pos;
  const move_t* moves = FWD ? ws.f_move : ws.r_move;
  u32* coll = FWD ? ws.f_coll : ws.r_coll;
  u32* barpos = stalling_barrier_positions<FWD>(ws);
  // positions of the compacted barriers and their indices (| HITBAR_HARD): at most BAR_FILL entries,
  // with a sentinel next to them so that the fixed-step searches need no range test -- rev: a word
  // of all ones behind the last entry (cp[cnt]; a step that overshoots reads it through one
  // `v_min` on the index); fwd: a zero in front of the first entry (the window starts one word into
  // the buffer; a step that undershoots reads cp[-1] through one `v_max`)
  constexpr u32 BAR_FILL = BAR_WIN - 1;
  u32* cp = reinterpret_cast<u32*>(c.lds.sort_lds) + (FWD ? 1 : 0);
  u32* ci = cp + BAR_WIN;
  const u32 j_rev0 = bc.n5 == 0 ? 0 : bc.n5 - 1;
  const u32 j_fwd0 = bc.n3 == 0 ? n - 1 : n - bc.n3;
  u32 carry_pos = 0;
  const u32 nh = wave::uniform(c.n_hit[FWD ? 1 : 0]);
  if (nh == 0 || n == 0) return 0;  // no barrier can stall a unit of this direction in this epoch
  // TRIALS: the listed units (ws.tmp[0]: rank, tmp[1]: list entry, tmp[2]: key; all three idle here)
  u32* const unit_rank = ws.tmp[0];
  u32* const unit_entry = ws.tmp[1];
  u32* const unit_key = ws.tmp[2];
  u32 n_listed = 0;
  const u32* hpos = ws.hit_pos[FWD ? 1 : 0];
  const u32* hidx = ws.hit_idx[FWD ? 1 : 0];
  constexpr u32 c0 = 0;
  u32 g0 = 0, g1 = 0, cnt = 0;      // the window holds list entries [g0, g1): cp[0 .. cnt)
  bool staged = false;
  u32 lo_cover = 1, hi_cover = 0;   // nothing staged yet (0xFFFFFFFF: no bound)
  u32 anchor = 0;                   // rev: entries below it lie before the batch; fwd: entries at
                                    // or above it lie beyond the batch (relative to c0)
  // Four consecutive ranks per lane, blocks of 256 ranks on 256-rank boundaries (128-bit loads);
  // rev: ranks ascending from j_rev0, fwd: ranks descending from j_fwd0 (lane 0 holds the highest
  // ranks of a block and walks its four units downwards).  Ranks outside the sweep are masked.
  const u32 b_first = (FWD ? j_fwd0 : j_rev0) / 256;
  const u32 nblk = FWD ? b_first + 1 : (n + 255) / 256 - b_first;
  const auto word0 = [&](u32 t) { return (FWD ? b_first - t : b_first + t) * 256 + 4 * (FWD ? 63 - lane : lane); };
  struct Blk {
    wave::U32x4 P, M;
  };
  const auto load_blk = [&](u32 t, Blk& r) {
    const u32 w = word0(t);
    const u32 wq = w < n ? w : 0u;
    r.P = wave::ld4(pos, wq);
    r.M = wave::ld4(moves, wq);
  };
  Blk cur;
  load_blk(0, cur);
  for (u32 t = 0; t < nblk; ++t) {
    const Blk g = cur;
    if (t + 1 < nblk) load_blk(t + 1, cur);
    const u32 w = word0(t);
    u32 k[4], P[4], lo_key[4], hi_key[4];
    bool bnd[4];
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {  // j: position in sweep order inside the lane
      const u32 q = FWD ? 3 - j : j;
      k[j] = w + q;
      const bool act = FWD ? k[j] <= j_fwd0 : (k[j] >= j_rev0 && k[j] < n);
      P[j] = act ? g.P.v[q] : 0u;
      bnd[j] = act && P[j] != UNBOUND;
    }
    const u32 nbr_in = wave::shfl_up1(P[3]);
    const u32 nbr0 = lane > 0 ? nbr_in : carry_pos;
    carry_pos = wave::bcast(P[3], 63);
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      const u32 q = FWD ? 3 - j : j;
      const u32 M = g.M.v[q];
      const bool first = k[j] == (FWD ? j_fwd0 : j_rev0);
      const u32 nbr = j == 0 ? nbr0 : P[j - 1];
      // see detect_lef_bar; 32-bit keys: positions lie below 2^32 - 2 (the host rejects longer
      // intervals), and a reach beyond that is as good as 2^32 - 2
      // (units that take no part: keys no window entry compares with -- nothing lies below 0,
      // nothing at or above 2^32 - 1)
      lo_key[j] = FWD ? 0xFFFFFFFFu : 0u;
      hi_key[j] = 0;
      if (bnd[j]) {
        if (!FWD) {
          const u32 reach = P[j] - M;
          lo_key[j] = first ? reach : umax(reach, nbr);
          hi_key[j] = P[j];
        } else {
          const u32 sum = P[j] + M;
          const u32 reach = (sum < P[j] || sum > 0xFFFFFFFEu) ? 0xFFFFFFFEu : sum;
          lo_key[j] = P[j] + 1;
          hi_key[j] = (first ? reach : umin(reach, nbr)) + 1;
        }
      }
    }
    const u64 bm = wave::ballot(bnd[0] || bnd[1] || bnd[2] || bnd[3]);
    if (bm == 0) continue;
    const u32 l_first = static_cast<u32>(wave::ctz64(bm));
    const u32 l_last = static_cast<u32>(63 - wave::clz64(bm));
    // keys of the lane's first / last bound unit in sweep order
    const u32 jf = bnd[0] ? 0u : bnd[1] ? 1u : bnd[2] ? 2u : 3u;
    const u32 jl = bnd[3] ? 3u : bnd[2] ? 2u : bnd[1] ? 1u : 0u;
    const u32 lo_f = jf == 0 ? lo_key[0] : jf == 1 ? lo_key[1] : jf == 2 ? lo_key[2] : lo_key[3];
    const u32 hi_f = jf == 0 ? hi_key[0] : jf == 1 ? hi_key[1] : jf == 2 ? hi_key[2] : hi_key[3];
    const u32 lo_l = jl == 3 ? lo_key[3] : jl == 2 ? lo_key[2] : jl == 1 ? lo_key[1] : lo_key[0];
    const u32 hi_l = jl == 3 ? hi_key[3] : jl == 2 ? hi_key[2] : jl == 1 ? hi_key[1] : hi_key[0];
    // keys the block spans (sweep order holds ascending positions for rev, descending for fwd)
    const u32 need_lo = FWD ? wave::bcast(lo_l, l_last) : wave::bcast(lo_f, l_first);
    const u32 need_hi = FWD ? wave::bcast(hi_f, l_first) : wave::bcast(hi_l, l_last);
#ifdef MODLE_SUBTIMER_LEFBAR
    const u64 t_stage = wave::clock();
#endif
    if (need_lo < lo_cover || need_hi > hi_cover) {
      // Move the window along the list to where this block starts (one coalesced load of
      // positions and indices).  Entries the window has already passed are dropped by counting;
      // when the block lies beyond the whole window, the window keeps moving.
      u32 moved = 0;
      for (;;) {
        if (staged) {
          // window entries before the block (rev: below need_lo; fwd: below need_hi)
          const u32 key = FWD ? need_hi : need_lo;
          u32 below = 0;
#pragma unroll
          for (u32 e0 = 0; e0 < BAR_WIN / 64; ++e0) {
            const u32 e = lane + 64 * e0;
            const u32 ce = cp[e];  // (e < BAR_WIN: inside the window whatever cnt is)
            below += static_cast<u32>(wave::popc64(wave::ballot((e < cnt) & (ce < key))));
          }
          if (!FWD) {
            g0 += below;
          } else {
            g1 = g0 + below;
          }
        } else {
          if (FWD) g1 = nh; else g0 = 0;
        }
        if (!FWD) {
          g1 = umin(g0 + BAR_FILL, nh);
        } else {
          g0 = g1 > BAR_FILL ? g1 - BAR_FILL : 0;
        }
        cnt = g1 - g0;
        wave::lockstep();
        {
          // what the window does not hold: everything before it lies below lo_cover,
          // everything after it at or above hi_cover
          const u32 edge_lo = g0 > 0 ? hpos[g0 - 1] : 0;
          const u32 edge_hi = g1 < nh ? hpos[g1] : 0;
          stage_stalling_window_call((MODLE_LDS u32*)cp, (MODLE_LDS u32*)ci, hpos + g0, hidx + g0, cnt);
          if (lane == 0) {
            if (FWD) cp[-1] = 0u; else cp[cnt] = 0xFFFFFFFFu;  // (the sentinel of the searches)
          }
          lo_cover = g0 > 0 ? wave::uniform(edge_lo) + 1 : 0;
          hi_cover = g1 < nh ? wave::uniform(edge_hi) : 0xFFFFFFFFu;
        }
        wave::sync_lds();
        staged = true;
        anchor = FWD ? cnt : 0;
        // done unless the block starts beyond this window and the list goes on
        const bool beyond = FWD ? (need_hi <= lo_cover && g0 > 0) : (need_lo >= hi_cover && g1 < nh);
        if (!beyond || ++moved > 64) break;  // (a block that is still not covered is looked up in device memory)
      }
    }
#ifdef MODLE_SUBTIMER_LEFBAR
    c.ph[14] += wave::clock() - t_stage;
    const u64 t_search = wave::clock();
#endif
    // per unit: the list entry at the near end of its window (rev: one past the last entry below the
    // unit; fwd: the first entry above it), when the window holds an entry at all
    u32 entry[4];
    bool has[4];
    u32 winner[4], bpos[4];  // (!TRIALS: the stalling barrier and its position)
    bool hard[4];
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      entry[j] = 0;
      has[j] = false;
      winner[j] = 0xFFFFFFFFu;
      bpos[j] = 0;
      hard[j] = false;
    }
    if (need_lo >= lo_cover && need_hi <= hi_cover) {
      // four searches side by side, all from the anchor the previous block left
      u32 q[4];
#pragma unroll
      for (u32 j = 0; j < 4; ++j) q[j] = anchor;
      if (!FWD) {
        // q = number of entries before the unit: the last of them is the candidate
#pragma unroll
        for (u32 sft = 64; sft >= 1; sft >>= 1) {
          // (the four reads of a round are issued together: left alone the compiler waits for each)
          u32 jx[4], kv[4];
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            jx[j] = q[j] + sft;
            kv[j] = (cp - 1)[umin(jx[j], cnt + 1)];  // (beyond the window: the sentinel)
          }
          wave::sched_fence();
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            if (kv[j] < hi_key[j]) q[j] = jx[j];
          }
          wave::sched_fence();
        }
        bool far = false;  // the fixed steps ran out: finish with a binary search (rare)
#pragma unroll
        for (u32 j = 0; j < 4; ++j) far = far || (bnd[j] && q[j] == anchor + HITBAR_NEAR && q[j] < cnt);
        if (wave::any(far)) {
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            if (bnd[j] && q[j] == anchor + HITBAR_NEAR && q[j] < cnt) {
              u32 hi = cnt;
              u32 l = q[j];
              while (l < hi) {
                const u32 mid = (l + hi) >> 1;
                if (cp[c0 + mid] < hi_key[j]) l = mid + 1; else hi = mid;
              }
              q[j] = l;
            }
          }
        }
        // the candidates of the four units: position and index read together, then tested
        u32 bp[4], wd[4];
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          const u32 e = c0 + (q[j] > 0 ? q[j] - 1 : 0);
          bp[j] = cp[e];
          wd[j] = ci[e];
        }
        wave::sched_fence();
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          if (bnd[j] & (q[j] > 0) & (bp[j] >= lo_key[j])) {
            has[j] = true;
            entry[j] = g0 + q[j];
            winner[j] = wd[j] & ~HITBAR_HARD;
            hard[j] = (wd[j] & HITBAR_HARD) != 0;
            bpos[j] = bp[j];
          }
        }
      } else {
        // q = number of entries at or before the unit: entry q is the candidate
#pragma unroll
        for (u32 sft = 64; sft >= 1; sft >>= 1) {
          u32 kv[4];
          i32 tq[4];
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            tq[j] = static_cast<i32>(q[j]) - static_cast<i32>(sft);
            kv[j] = cp[tq[j] > -1 ? tq[j] : -1];  // (before the window: the sentinel)
          }
          wave::sched_fence();
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            if (kv[j] >= lo_key[j]) q[j] = static_cast<u32>(tq[j]);
          }
          wave::sched_fence();
        }
        bool far = false;  // the fixed steps ran out: finish with a binary search (rare)
#pragma unroll
        for (u32 j = 0; j < 4; ++j) far = far || (bnd[j] && q[j] + HITBAR_NEAR == anchor && q[j] > 0);
        if (wave::any(far)) {
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            if (bnd[j] && q[j] + HITBAR_NEAR == anchor && q[j] > 0) {
              u32 lo = 0;
              u32 h = q[j];
              while (lo < h) {
                const u32 mid = (lo + h) >> 1;
                if (cp[c0 + mid] < lo_key[j]) lo = mid + 1; else h = mid;
              }
              q[j] = h;
            }
          }
        }
        // the candidates of the four units: position and index read together, then tested
        u32 bp[4], wd[4];
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          const u32 e = c0 + (q[j] < cnt ? q[j] : 0);
          bp[j] = cp[e];
          wd[j] = ci[e];
        }
        wave::sched_fence();
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          if (bnd[j] & (q[j] < cnt) & (bp[j] < hi_key[j])) {
            has[j] = true;
            entry[j] = g0 + q[j];
            winner[j] = wd[j] & ~HITBAR_HARD;
            hard[j] = (wd[j] & HITBAR_HARD) != 0;
            bpos[j] = bp[j];
          }
        }
      }
      const u32 q_last = jl == 3 ? q[3] : jl == 2 ? q[2] : jl == 1 ? q[1] : q[0];
      anchor = wave::bcast(q_last, l_last);
    } else {
      // the block spans more list entries than the window holds (few, far apart units): per-unit
      // binary searches in the list in device memory
#pragma unroll
      for (u32 j = 0; j < 4; ++j) {
        if (bnd[j]) {
          if (!FWD) {
            const u32 qa = lower_bound_u32(hpos, nh, hi_key[j]);  // entries below the unit
            if (qa > 0) {
              const u32 bp_ = hpos[qa - 1];
              if (bp_ >= lo_key[j]) {
                const u32 wd_ = hidx[qa - 1];
                has[j] = true;
                entry[j] = qa;
                winner[j] = wd_ & ~HITBAR_HARD;
                hard[j] = (wd_ & HITBAR_HARD) != 0;
                bpos[j] = bp_;
              }
            }
          } else {
            const u32 qa = lower_bound_u32(hpos, nh, lo_key[j]);  // first entry above the unit
            if (qa < nh) {
              const u32 bp_ = hpos[qa];
              if (bp_ < hi_key[j]) {
                const u32 wd_ = hidx[qa];
                has[j] = true;
                entry[j] = qa;
                winner[j] = wd_ & ~HITBAR_HARD;
                hard[j] = (wd_ & HITBAR_HARD) != 0;
                bpos[j] = bp_;
              }
            }
          }
        }
        // (the loads of this rare path end here: see detect_primary)
        wave::pin(entry[j]);
        wave::pin(winner[j]);
        wave::pin(bpos[j]);
        u32 hd = (hard[j] ? 1u : 0u) | (has[j] ? 2u : 0u);
        wave::pin(hd);
        hard[j] = (hd & 1u) != 0;
        has[j] = (hd & 2u) != 0;
      }
      // the staged entries stay valid, but the next block must not trust the anchor
      lo_cover = 1;
      hi_cover = 0;
    }
#ifdef MODLE_SUBTIMER_LEFBAR
    c.ph[15] += wave::clock() - t_search;
#endif
    if (wave::any(has[0] || has[1] || has[2] || has[3])) {
      if (!TRIALS) {
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          if (has[j]) {
            coll[k[j]] = cw_make(winner[j], EV_COLLISION | EV_LEF_BAR) | (hard[j] ? CW_HARD : 0u);
            barpos[k[j]] = bpos[j];
          }
        }
      } else {
        // listed in sweep order: lane by lane, the four units of a lane in turn
        u32 before[4];
        u32 lane_cnt = 0;
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          before[j] = lane_cnt;
          lane_cnt += has[j] ? 1u : 0u;
        }
        const u32 ps = wave_prefix_sum_u32(lane_cnt);
        const u32 first = n_listed + ps - lane_cnt;
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          if (has[j]) {
            unit_rank[first + before[j]] = k[j];
            unit_entry[first + before[j]] = entry[j];
            unit_key[first + before[j]] = FWD ? hi_key[j] : lo_key[j];
          }
        }
        n_listed += wave::bcast(ps, 63);
      }
    }
  }
  wave::sync_mem();
  return n_listed;
}

// Second pass with Bernoulli trials: 64 listed units at a time, in list order (rev: ascending
// ranks, fwd: descending ranks -- the order in which the reference reaches their barriers).  A
// unit's window is the run of list entries [e_lo, e_hi) between the entry the sweep found and its
// far key; the reference visits them in ascending order for a rev unit and in descending order for
// a fwd unit, draws for every entry whose probability is below one, and keeps the last hit.
template <bool FWD>
MODLE_DEV_NOINLINE void resolve_listed_units(Cell& c, u32 n_listed) {
  Workspace& ws = c.ws;
  const Params& p = *c.p;
  const u32 lane = wave::lane();
  const u32 nh = wave::uniform(c.n_hit[FWD ? 1 : 0]);
  const u32* hpos = ws.hit_pos[FWD ? 1 : 0];
  const u32* hidx = ws.hit_idx[FWD ? 1 : 0];
  const u32* unit_rank = ws.tmp[0];
  const u32* unit_entry = ws.tmp[1];
  const u32* unit_key = ws.tmp[2];
  u32* coll = FWD ? ws.f_coll : ws.r_coll;
  u32* barpos = stalling_barrier_positions<FWD>(ws);
  const f64 pb_major = wave::own_regs(p.pblock_major), pb_minor = wave::own_regs(p.pblock_minor);
  const bool major_trial = pb_major != 1.0, minor_trial = pb_minor != 1.0;  // (neither is zero on a list)
  // entries of a window in the reference's visiting order: q = 0 .. e_hi - e_lo - 1
  const auto visit = [&](u32 e_lo, u32 e_hi, u32 q) { return FWD ? e_hi - 1 - q : e_lo + q; };
  for (u32 base = 0; base < n_listed; base += 64) {
    const u32 u = base + lane;
    const bool act = u < n_listed;
    const u32 k = wave::ld_sel(unit_rank, u, act, 0u);
    const u32 e0 = wave::ld_sel(unit_entry, u, act, 0u);
    const u32 key = wave::ld_sel(unit_key, u, act, 0u);
    // the far end of the window (a run of one entry nearly always)
    u32 e_lo = e0, e_hi = e0;
    if (act) {
      if (!FWD) {
        while (e_lo > 0 && hpos[e_lo - 1] >= key) --e_lo;  // (key: the window's lower bound)
      } else {
        while (e_hi < nh && hpos[e_hi] < key) ++e_hi;      // (key: one past its upper bound)
      }
    }
    // trials this unit consumes
    u32 ntr = 0;
    for (u32 q = 0; q < e_hi - e_lo; ++q) {
      const bool is_major = (hidx[visit(e_lo, e_hi, q)] & HITBAR_HARD) != 0;
      ntr += (is_major ? major_trial : minor_trial) ? 1u : 0u;
    }
    u32 off = wave_prefix_sum_u32(ntr);
    const u32 total = wave::bcast(off, 63);
    off -= ntr;
    // winner of a unit whose trials start `first` outputs into the stream from g.pos
    const auto pick = [&](u32 first, u32& w_entry) {
      u32 t = 0;
      w_entry = 0xFFFFFFFFu;
      for (u32 q = 0; q < e_hi - e_lo; ++q) {
        const u32 e = visit(e_lo, e_hi, q);
        const bool is_major = (hidx[e] & HITBAR_HARD) != 0;
        bool hit = true;
        if (is_major ? major_trial : minor_trial) {
          hit = bernoulli_raw(rng_peek(c.g, c.g.pos + first + t), is_major ? pb_major : pb_minor);
          ++t;
        }
        if (hit) w_entry = e;  // later visits overwrite earlier ones
      }
    };
#ifdef MODLE_EMU_TRACE_RANK  // (emulator only: which regime a test exercises)
    if (lane == 0) fprintf(stderr, "lef_bar_trials: %s batch of %u units, %u trials%s\n", FWD ? "fwd" : "rev",
                           umin(64u, n_listed - base), total, total > RNG_BLOCK ? " (resolved in rounds)" : "");
#endif
    u32 w_entry = 0xFFFFFFFFu;
    if (total <= RNG_BLOCK) {
      if (total != 0) rng_ensure(c.g, total);
      if (act) pick(off, w_entry);
      rng_advance(c.g, total);
    } else {
      // More trials in this batch than one block of the PRNG ring serves (dense barrier
      // annotations): the lanes are resolved in rounds, each taking the longest run of lanes (in
      // lane = stream order) whose trials fit one block; a single unit with more trials than that is
      // replayed sequentially.
      u64 pend = wave::ballot(act);
      u32 base_tr = 0;  // trials consumed by the lanes resolved so far
      while (pend != 0) {
        const bool mine_pending = ((pend >> lane) & 1u) != 0;
        const bool fits = mine_pending && (off + ntr - base_tr <= RNG_BLOCK);
        const u64 fm = wave::ballot(fits);
        if (fm == 0) {
          const u32 l = static_cast<u32>(wave::ctz64(pend));
          const u32 lo = wave::bcast(e_lo, l), hi = wave::bcast(e_hi, l);
          u32 w = 0xFFFFFFFFu;
          for (u32 q = 0; q < hi - lo; ++q) {
            const u32 e = FWD ? hi - 1 - q : lo + q;
            const bool is_major = (wave::uniform(hidx[e]) & HITBAR_HARD) != 0;
            bool hit = true;
            if (is_major ? major_trial : minor_trial) hit = bernoulli_raw(rng_next(c.g), is_major ? pb_major : pb_minor);
            if (hit) w = e;
          }
          if (lane == l) w_entry = w;
          base_tr += wave::bcast(ntr, l);
          pend &= ~(u64(1) << l);
        } else {
          // fitting lanes are a run of pending lanes starting at the first one
          const u32 l_last_fit = static_cast<u32>(63 - wave::clz64(fm));
          const u32 cnt = wave::bcast(off + ntr, l_last_fit) - base_tr;
          if (cnt != 0) rng_ensure(c.g, cnt);
          if (fits) pick(off - base_tr, w_entry);
          rng_advance(c.g, cnt);
          base_tr += cnt;
          pend &= ~fm;
        }
      }
    }
    if (w_entry != 0xFFFFFFFFu) {
      const u32 wd = hidx[w_entry];
      coll[k] = cw_make(wd & ~HITBAR_HARD, EV_COLLISION | EV_LEF_BAR) | ((wd & HITBAR_HARD) ? CW_HARD : 0u);
      barpos[k] = hpos[w_entry];
    }
  }
  wave::sync_mem();
}

template <bool FWD>
MODLE_DEV_NOINLINE void detect_lef_bar(Cell& c, BoundaryCounts bc) {
  if (wave::uniform(c.iv->n_barriers) == 0) return;
  if (!lef_bar_trials_needed(*c.p)) {
    detect_lef_bar_det<FWD>(c, bc);
    return;
  }
  const u32 n_listed = detect_lef_bar_sweep<FWD, true>(c, bc);
  if (n_listed != 0) resolve_listed_units<FWD>(c, n_listed);
}

// compute_lef_lef_collision_pos (reference: simulation.cpp:523-551)
MODLE_DEV void lef_lef_collision_pos(u32 rev_p, u32 fwd_p, u32 rev_move, u32 fwd_move,
                                     u32& out_rev, u32& out_fwd) {
  // (all operands are below 2^32: the sum of the two converted moves is exact and equals the
  // converted 64-bit sum, and the rounded product is at most fwd_move: 32-bit conversions)
  const f64 relative_speed = static_cast<f64>(rev_move) + static_cast<f64>(fwd_move);
  const f64 ttc = static_cast<f64>(rev_p - fwd_p) / relative_speed;
  const u32 cpos = fwd_p + static_cast<u32>(wave::f_round(static_cast<f64>(fwd_move) * ttc));
  if (cpos == fwd_p) {
    out_rev = cpos + 1;
    out_fwd = cpos;
  } else {
    out_rev = cpos;
    out_fwd = cpos - 1;
  }
}

// Position of the barrier a stalled unit's collision word points at.  The reference indexes the
// barrier array with the word's index without checking that the word is a LEF-BAR collision
// (simulation_detect_collisions.cpp:371, 389; only asserted in debug builds): a unit flagged at
// the interval boundary (index 5 / 3) that still takes part in the primary pass makes it read
// barrier #5 / #3, or past the end of the array when there are fewer barriers.  In-range
// indices behave like the reference; out-of-range ones (undefined behaviour there) read as 0.
MODLE_DEV u32 stalling_barrier_pos(const Interval& iv, u32 word) {
  const u32 idx = cw_index(word);
  return idx < iv.n_barriers ? iv.bar_pos[idx] : 0u;
}

// detect_primary_lef_lef_collisions (reference: simulation_detect_collisions.cpp:250-397),
// evaluated per rev unit: the merge loop pairs the rev unit of rank j with the last fwd unit
// strictly upstream of it, provided j is the first rev unit downstream of that fwd unit and the
// fwd unit is not the last one the loop is allowed to look at.
//
// With `fuse_correct`, correct_moves_for_primary_lef_lef_collisions (reference:
// simulation_correct_moves.cpp:53-121) is applied on the spot: every unit takes part in at most
// one pair and the pair's corrected moves depend only on the two units (original moves, or
// "distance to the stalling barrier - 1" for a unit that stays stalled by a barrier, which is
// what correct_moves_for_lef_bar_collisions stores for it).
// Two passes (round 3).  A rev unit pairs with the fwd unit right upstream of it only when the two
// can meet within this epoch's moves, which a few per cent of the units can; the one-pass form
// nevertheless loaded five words per rev unit and staged five 256-entry slices of the fwd side in
// LDS per block of 128 rev units.  Pass 1 reads positions and rev moves only: it finds every rev
// unit's partner (a search in a staged slice of fwd POSITIONS) and lists the pairs that pass the
// geometric test with the largest fwd move of the epoch in place of the partner's own move (the
// move adjustment reports it: a superset).  Pass 2 takes 64 listed pairs at a time, gathers both
// units' words and decides, draws and corrects exactly as the reference's merge loop does; list
// order = rank order = draw order, and no unit belongs to two pairs.
#ifndef MODLE_PRIMARY_BACK
#define MODLE_PRIMARY_BACK 128  // (a smaller value in a test build of the emulator exercises the fall-back)
#endif
constexpr u32 PRIMARY_BACK = MODLE_PRIMARY_BACK;  // fwd ranks before a block's first rev rank in its slice
static_assert(PRIMARY_BACK + 128 <= STAGE_CAP, "the slice must reach the block's last rank");
struct PrimaryBatch {
  wave::U32x2 R, rev_move;
  u32 sp[STAGE_CAP / 64];
};
// `base` is even; ranks outside [first, n) are masked where the values are used
template <class Op>
MODLE_DEV void primary_load_batch(Op op, const Workspace& ws, u32 n, u32 base, u32 w0, u32 lane,
                                  PrimaryBatch& b, bool rev_side) {
  if (rev_side) {
    const u32 k0 = base + 2 * lane;
    const u32 kq = k0 < n ? k0 : 0u;
    b.R = wave::ld2(ws.r_pos, kq);
    b.rev_move = wave::ld2(ws.r_move, kq);
  }
#pragma unroll
  for (u32 q = 0; q < STAGE_CAP / 64; ++q) {
    const u32 t = lane + 64 * q;
    b.sp[q] = op(ws.f_pos, w0 + t, w0 + t < n, UNBOUND, b.sp[q]);
  }
}

MODLE_DEV_NOINLINE void detect_primary(Cell& c, BoundaryCounts bc, bool fuse_correct) {
  Workspace& ws = c.ws;
  const Params& p = *c.p;
  const Interval& iv = *c.iv;
  const u32 n = wave::uniform(c.n_active);
  if (bc.n5 == n || bc.n3 == n) return;
  const u32 lane = wave::lane();
  const u32 i2 = bc.n3 == 0 ? n : n - (bc.n3 - 1);
  // run_lef_lef_collision_trial (simulation_impl.hpp:93-96): no draw when the bypass probability
  // is 0 (always collide) -- and none when it is 1: bernoulli_distribution(0) returns false
  // without touching the engine
  const f64 p_collide = 1.0 - p.p_bypass;
  const bool never_collide = p.p_bypass != 0.0 && p_collide == 0.0;
  const bool trials = p.p_bypass != 0.0 && !never_collide;
  const u32 prim = EV_COLLISION | EV_LEF_LEF_PRIMARY;
  u32* stage = c.lds.stage;  // slice of the fwd positions, ranks [w0, w0 + STAGE_CAP)
  // the list of pairs (rank of the rev unit, fwd units strictly upstream of it) in scratch that is
  // idle until the secondary pass; lanes with nothing to store hit scratch words of their own
  u32* const q_k = ws.tmp[0];
  u32* const q_pf = ws.tmp[1];
  u32* const dump = reinterpret_cast<u32*>(ws.sort_keys) + 2 * lane;
  const u64 fwd_reach = wave::uniform(c.max_fwd_move);
  u32 n_cand = 0;
  u32 carry_pos = 0;
  // ---- pass 1 ----------------------------------------------------------------------------------
  // pf = number of fwd units strictly upstream of R.  Every LEF has its rev unit at or upstream of
  // its fwd unit, so pf(k) = k - (LEFs whose loop spans the position of rev unit k): at most k, and
  // below it by the local depth of coverage, a few units to a few dozen.  The slice of fwd positions
  // a block of 128 rev ranks searches is therefore the FIXED window of ranks [base - PRIMARY_BACK,
  // base + 128): nothing about the loads of a block depends on the search of the block before it
  // (it used to: the slice started where the previous block's last unit fell, so every block paid
  // a full memory round trip after its search), and they are requested two blocks ahead.  A unit
  // with more than PRIMARY_BACK loops over it is looked up in device memory.
  const u32 first = bc.n5;
  const auto slice_start = [](u32 base) { return base > PRIMARY_BACK ? base - PRIMARY_BACK : 0u; };
  {
    const u32 b0 = first & ~1u;
    PrimaryBatch pa, pb;
    primary_load_batch(wave::LdRaw{}, ws, n, b0, slice_start(b0), lane, pa, true);
    if (b0 + 128 < n) primary_load_batch(wave::LdRaw{}, ws, n, b0 + 128, slice_start(b0 + 128), lane, pb, true);
    const auto block = [&](PrimaryBatch& cur, u32 base) {
      const u32 w0 = slice_start(base);
      primary_load_batch(wave::LdMask{}, ws, n, base, w0, lane, cur, false);  // (defaults outside the range)
      u32 k[2], R[2], rev_move[2];
      bool act[2];
#pragma unroll
      for (u32 j = 0; j < 2; ++j) {
        k[j] = base + 2 * lane + j;
        act[j] = k[j] >= first && k[j] < n;
        R[j] = act[j] ? cur.R.v[j] : UNBOUND;
        rev_move[j] = act[j] ? cur.rev_move.v[j] : 0u;
      }
      wave::lockstep();
#pragma unroll
      for (u32 q = 0; q < STAGE_CAP / 64; ++q) stage[lane + 64 * q] = cur.sp[q];
      wave::sync_lds();
      // (the block's registers are free: request the block after the next one)
      if (base + 256 < n) primary_load_batch(wave::LdRaw{}, ws, n, base + 256, slice_start(base + 256), lane, cur, true);
      const u32 prev_in = wave::shfl_up1(R[1]);
      const u32 Rprev0 = lane > 0 ? prev_in : carry_pos;
      // number of staged positions below R: a fixed-step search (no loop control, the eight steps
      // are the same for every unit; the two reads of a round are issued together)
      u32 lo[2] = {0, 0};
      static_assert(STAGE_CAP == 256, "the search below covers 256 entries");
#pragma unroll
      for (u32 sft = 128; sft >= 1; sft >>= 1) {
        u32 sv[2];
#pragma unroll
        for (u32 j = 0; j < 2; ++j) sv[j] = stage[lo[j] + sft - 1];
        wave::sched_fence();
#pragma unroll
        for (u32 j = 0; j < 2; ++j) {
          if (sv[j] < R[j]) lo[j] += sft;
        }
        wave::sched_fence();
      }
      const u32 st_last = stage[STAGE_CAP - 1];
      u32 pf[2] = {0, 0};
      bool beyond = false;  // the answer lies outside the slice (rare): device memory
#pragma unroll
      for (u32 j = 0; j < 2; ++j) {
        u32 l = lo[j];
        if (l == STAGE_CAP - 1 && st_last < R[j]) l = STAGE_CAP;
        pf[j] = umin(w0 + l, n);
        // every staged position is at or above R and the slice does not start at rank 0: fwd units
        // before the slice may be at or above R too.  (The other end cannot be exceeded: pf <= k.)
        beyond = beyond || (act[j] && ((l == 0 && w0 > 0) || (l == STAGE_CAP && w0 + STAGE_CAP < n)));
      }
      if (wave::any(beyond)) {
#pragma unroll
        for (u32 j = 0; j < 2; ++j) {
          const u32 l = pf[j] - w0;
          if (act[j] && ((l == 0 && w0 > 0) || (l == STAGE_CAP && w0 + STAGE_CAP < n)))
            pf[j] = lower_bound_u32(ws.f_pos, n, R[j]);
          wave::pin(pf[j]);  // (the loads of this rare path end here)
        }
      }
      // the partner of each unit (the fwd unit right upstream of it): its position from the slice,
      // without branches; partners beyond the slice come from device memory
      u32 F[2];
      bool has[2], staged[2];
#pragma unroll
      for (u32 j = 0; j < 2; ++j) {
        has[j] = act[j] && pf[j] >= 1 && pf[j] < i2;
        const u32 kf = pf[j] - 1;
        staged[j] = has[j] && kf >= w0 && kf - w0 < STAGE_CAP;
        F[j] = stage[staged[j] ? kf - w0 : 0u];
      }
      wave::sched_fence();
      if (wave::any((has[0] && !staged[0]) || (has[1] && !staged[1]))) {
#pragma unroll
        for (u32 j = 0; j < 2; ++j) {
          if (has[j] && !staged[j]) F[j] = ws.f_pos[pf[j] - 1];
          // (the load ends HERE: where a value loaded on a rare path merges with the common path
          // the compiler waits for everything in flight -- the next block's loads -- on both)
          wave::pin(F[j]);
        }
      }
      bool cand[2];
#pragma unroll
      for (u32 j = 0; j < 2; ++j) {
        const u32 Rprev = j == 0 ? Rprev0 : R[0];
        const bool first_after = (k[j] == first) || Rprev <= F[j];
        const u32 delta = R[j] - F[j];  // > 0 by construction (where it is used)
        cand[j] = has[j] && first_after && static_cast<u64>(delta) < static_cast<u64>(rev_move[j]) + fwd_reach;
      }
      const u64 cm0 = wave::ballot(cand[0]), cm1 = wave::ballot(cand[1]);
      {
        // rank order: unit (lane, j) after the units of the lanes before it and after unit 0 of its
        // own lane
        const u64 lt = lanemask_lt(lane);
        const u32 e0 = n_cand + static_cast<u32>(wave::popc64(cm0 & lt) + wave::popc64(cm1 & lt));
        const u32 e1 = e0 + (cand[0] ? 1u : 0u);
        *(cand[0] ? &q_k[e0] : dump) = k[0];
        *(cand[0] ? &q_pf[e0] : dump + 1) = pf[0];
        *(cand[1] ? &q_k[e1] : dump) = k[1];
        *(cand[1] ? &q_pf[e1] : dump + 1) = pf[1];
      }
      n_cand += static_cast<u32>(wave::popc64(cm0) + wave::popc64(cm1));
      carry_pos = wave::bcast(R[1], 63);
    };
    for (u32 base = b0; base < n; base += 256) {
      block(pa, base);
      if (base + 128 < n) block(pb, base + 128);
    }
  }
  wave::sync_mem();
  // ---- pass 2 ----------------------------------------------------------------------------------
  struct PairRegs {
    u32 k, pf, R, rev_move, rev_id, rc, rbp, F, fwd_move, fwd_id, fc, fbp;
  };
  const auto load_pairs = [&](u32 base, u32 kk, u32 pp, PairRegs& r) {
    const bool in = base + lane < n_cand;
    const u32 k = in ? kk : 0u, kf = in ? pp - 1 : 0u;  // (a listed pair has pf >= 1)
    r.k = kk;
    r.pf = pp;
    r.R = wave::LdRaw{}(ws.r_pos, k, true, 0, 0u);
    r.rev_move = wave::LdRaw{}(ws.r_move, k, true, 0, 0u);
    r.rev_id = wave::LdRaw{}(ws.r_id, k, true, 0, 0u);
    r.rc = wave::LdRaw{}(ws.r_coll, k, true, 0, 0u);
    r.rbp = wave::LdRaw{}(stalling_barrier_positions<false>(ws), k, true, 0, 0u);
    r.F = wave::LdRaw{}(ws.f_pos, kf, true, 0, 0u);
    r.fwd_move = wave::LdRaw{}(ws.f_move, kf, true, 0, 0u);
    r.fwd_id = wave::LdRaw{}(ws.f_id, kf, true, 0, 0u);
    r.fc = wave::LdRaw{}(ws.f_coll, kf, true, 0, 0u);
    r.fbp = wave::LdRaw{}(stalling_barrier_positions<true>(ws), kf, true, 0, 0u);
  };
  const auto list_k = [&](u32 base) { return wave::ld_sel(q_k, base + lane, base + lane < n_cand, 0u); };
  const auto list_pf = [&](u32 base) { return wave::ld_sel(q_pf, base + lane, base + lane < n_cand, 1u); };
  PairRegs pcur;
  u32 nk = 0, np = 1;
  if (n_cand != 0) {
    load_pairs(0, list_k(0), list_pf(0), pcur);
    if (64 < n_cand) {
      nk = list_k(64);
      np = list_pf(64);
    }
  }
  for (u32 base = 0; base < n_cand; base += 64) {
    const PairRegs q = pcur;
    if (base + 64 < n_cand) {
      load_pairs(base + 64, nk, np, pcur);
      if (base + 128 < n_cand) {
        nk = list_k(base + 128);
        np = list_pf(base + 128);
      }
    }
    const bool valid = base + lane < n_cand;
    const u32 delta = q.R - q.F;
    // (pass 1 has checked that the pair exists and that the rev unit is the first one downstream of
    // the fwd unit; what is left is the geometric test with the partner's own move)
    const bool cand = valid && static_cast<u64>(delta) < static_cast<u64>(q.rev_move) + q.fwd_move;
    const u64 cm = wave::ballot(cand);
    bool hit = cand && !never_collide;
    if (trials && cm != 0) {
      const u32 cnt = static_cast<u32>(wave::popc64(cm));
      rng_ensure(c.g, cnt);
      const u32 t = static_cast<u32>(wave::popc64(cm & lanemask_lt(lane)));
      hit = cand && bernoulli_raw(rng_peek(c.g, c.g.pos + t), p_collide);
      rng_advance(c.g, cnt);
    }
    const auto handle_hit = [&](u32 pf_h, u32 k_h, u32 R_h, u32 F_h, u32 rev_move_h, u32 fwd_move_h, u32 rev_id_k_h,
                                u32 fwd_id_s_h, u32 rc_k_h, u32 fc_s_h, u32 rbp_k_h, u32 fbp_s_h) {
      const u32 kf = pf_h - 1;
      const u32 rev_id = rev_id_k_h, fwd_id = fwd_id_s_h;
      u32 cpos_rev, cpos_fwd;
      lef_lef_collision_pos(R_h, F_h, rev_move_h, fwd_move_h, cpos_rev, cpos_fwd);
      const u32 rc = rc_k_h, fc = fc_s_h;
      const bool rev_occ = cw_occurred(rc), fwd_occ = cw_occurred(fc);
      u32 rev_other = 0, fwd_other = 0;
      const bool rev_odd = rev_occ && !cw_occurred_as(rc, EV_LEF_BAR);
      const bool fwd_odd = fwd_occ && !cw_occurred_as(fc, EV_LEF_BAR);
      // a stalled unit whose word is not a LEF-BAR collision (flagged at the interval boundary):
      // the barrier its index points at is read on a path of its own, and the load ends there
      // (see above)
      if (rev_odd || fwd_odd) {
        if (rev_odd) rev_other = stalling_barrier_pos(iv, rc);
        if (fwd_odd) fwd_other = stalling_barrier_pos(iv, fc);
        wave::pin(rev_other);
        wave::pin(fwd_other);
      }
      bool both = false;
      if (!rev_occ && !fwd_occ) {
        ws.r_coll[k_h] = cw_make(fwd_id, prim);
        ws.f_coll[kf] = cw_make(rev_id, prim);
        both = true;
      } else if (rev_occ && !fwd_occ) {
        const u32 barrier_pos = rev_odd ? rev_other : rbp_k_h;
        ws.f_coll[kf] = cw_make(rev_id, prim);
        if (cpos_fwd > barrier_pos) {
          // the LEF-LEF collision happens before the predicted LEF-BAR one
          ws.r_coll[k_h] = cw_make(fwd_id, prim);
          both = true;
        } else if (fuse_correct && cw_occurred_as(rc, EV_LEF_BAR)) {
          // fwd unit runs into a rev unit that stays stalled 1 bp downstream of its barrier
          const u32 rev_move_stalled = (R_h - barrier_pos) - 1;
          ws.f_move[kf] = (R_h - rev_move_stalled) - F_h - 1;
        }
      } else if (!rev_occ && fwd_occ) {
        const u32 barrier_pos = fwd_odd ? fwd_other : fbp_s_h;
        ws.r_coll[k_h] = cw_make(fwd_id, prim);
        if (cpos_rev < barrier_pos) {
          ws.f_coll[kf] = cw_make(rev_id, prim);
          both = true;
        } else if (fuse_correct && cw_occurred_as(fc, EV_LEF_BAR)) {
          const u32 fwd_move_stalled = (barrier_pos - F_h) - 1;
          ws.r_move[k_h] = R_h - (F_h + fwd_move_stalled) - 1;
        }
      }
      if (both && fuse_correct) {
        ws.r_move[k_h] = R_h - cpos_rev;
        ws.f_move[kf] = cpos_fwd - F_h;
      }
    };
    if (wave::any(hit)) {
      if (hit) handle_hit(q.pf, q.k, q.R, q.F, q.rev_move, q.fwd_move, q.rev_id, q.fwd_id, q.rc, q.fc, q.rbp, q.fbp);
    }
  }
  wave::sync_mem();
}

// correct_moves_for_primary_lef_lef_collisions (reference: simulation_correct_moves.cpp:53-121)
// as a stand-alone pass; only used by the phase-level test entry point when the reference's
// hook sequence runs it separately from detection.
MODLE_DEV_NOINLINE void correct_moves_primary_standalone(Cell& c) {
  ensure_inverse_both(c);
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    if (k < n) {
      const u32 rc = ws.r_coll[k];
      if (cw_occurred_as(rc, EV_LEF_LEF_PRIMARY)) {
        const u32 kf = ws.f_rank[cw_index(rc)];
        const u32 fc = ws.f_coll[kf];
        if (cw_occurred_as(fc, EV_LEF_LEF_PRIMARY)) {
          u32 p1, p2;
          lef_lef_collision_pos(ws.r_pos[k], ws.f_pos[kf], ws.r_move[k], ws.f_move[kf], p1, p2);
          ws.r_move[k] = ws.r_pos[k] - p1;
          ws.f_move[kf] = p2 - ws.f_pos[kf];
        } else if (cw_occurred_as(fc, EV_LEF_BAR)) {
          ws.r_move[k] = ws.r_pos[k] - (ws.f_pos[kf] + ws.f_move[kf]) - 1;
        }
      }
    }
  }
  wave::sync_mem();
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    if (k < n) {
      const u32 fc = ws.f_coll[k];
      if (cw_occurred_as(fc, EV_LEF_LEF_PRIMARY)) {
        const u32 kr = ws.r_rank[cw_index(fc)];
        if (cw_occurred_as(ws.r_coll[kr], EV_LEF_BAR))
          ws.f_move[k] = (ws.r_pos[kr] - ws.r_move[kr]) - ws.f_pos[k] - 1;
      }
    }
  }
  wave::sync_mem();
}

// process_secondary_lef_lef_collisions (reference: simulation_detect_collisions.cpp:400-515).
// The pass is a chain: a stalled unit can stall its follower, which can stall the next one, and
// every candidate consumes one Bernoulli draw in rank order.  Ranks whose collision was avoided
// are appended to `list` (rank positions, visiting order) for fix_secondary.
//
// correct_moves_for_lef_bar_collisions (reference: simulation_correct_moves.cpp:19-50) is fused
// into the first pass: a unit stalled by a barrier gets move = distance - 1.  (It has to come after
// primary detection, which tests the uncorrected moves.)
//
// Two passes.  Nearly every batch of 64 consecutive ranks holds a few candidates (units
// queued behind a stalled unit try again in every epoch), so a one-pass form (rounds 1-2) ran its
// chain resolution -- a long dependent sequence of ballots, scalar bit operations, LDS reads and
// draws -- once per batch for a handful of useful lanes.  The first pass only corrects the LEF-BAR
// moves and FILTERS: the candidates (a superset: units that can reach their blocker's position and
// whose blocker is, or may become, stalled) are appended in visiting order to a compact list of
// ranks in device scratch.  The second pass resolves 64 CANDIDATES at a time: it gathers their
// units and their blockers (the unit of the adjacent rank) and runs the chain logic once for 64
// useful lanes.  A candidate whose blocker is a candidate too (`cont`) finds it in the lane before
// it (or in the carry of the previous group): runs of such lanes are the chains.  Draw order =
// list order = visiting order.  The Bernoulli outcomes of a group are evaluated up front for the
// first 64 outputs of the stream; which unit takes which output follows from masks, with one
// round per avoided collision that cuts a chain short (not per avoided collision).
constexpr u32 SEC_CONT = 0x40000000u;  // on the rank word of a list entry: the blocker is the entry before
constexpr u32 SEC_BOCC = 0x80000000u;  // ... the blocker (not a candidate) is stalled

// Pass 1 as a stepper, so that the rev and the fwd instance can share one loop (they are
// independent: no draws, each reads and writes its own direction's arrays): two dependency chains
// per iteration instead of one.
// composition of two steps of the candidate recurrence x' = p | (q & x), packed as p | q << 1:
// `later` applied after `earlier`
MODLE_DEV u32 sec_compose(u32 earlier, u32 later) {
  const u32 p = (later & 1u) | ((later >> 1) & earlier & 1u);
  const u32 q = (later >> 1) & (earlier >> 1) & 1u;
  return p | (q << 1);
}
MODLE_DEV u32 wave_prefix_sec_compose(u32 v) {
#define MODLE_STEP(S) v = sec_compose(wave::scan_move<S>(v, 2u), v);
  MODLE_SCAN_STEPS(MODLE_STEP)
#undef MODLE_STEP
  return v;
}

template <bool FWD>
struct SecondaryFilter {
  // FOUR consecutive ranks per lane, blocks of 256 ranks (128-bit loads); fwd: lane 0 holds the
  // highest ranks of a block and walks its four units downwards, so that (lane, unit) is the
  // visiting order in both directions.  "Is a candidate" is a recurrence along the visiting order,
  //     x[u] = pot[u] & (blocker_stalled[u] | x[u - 1])
  // (pot: free follower that can reach its blocker's position): a step is the function
  // x -> p | (q & x) with p = pot & blocker_stalled, q = pot, steps compose to functions of the
  // same form, so the lane composes its four steps, ONE cross-lane scan composes the lanes, and
  // every lane replays its four steps from the value that enters it.
  struct Blk {
    wave::U32x4 P, M, C, B;
  };
  const u32 *pos, *coll, *barpos;
  move_t *moves, *dump_m;
  u32 *q_k, *dump;
  u32 n, lane, nblk, cap, n_cand, carry_pos, carry_coll;
  i32 f_first;
  bool correct_lef_bar, do_secondary, carry_pending;
  Blk cur;

  // first rank of this lane in block t of the sweep (t = 0 is the block the sweep starts with)
  MODLE_DEV_MEMBER u32 word0(u32 t) const { return (FWD ? nblk - 1 - t : t) * 256 + 4 * (FWD ? 63 - lane : lane); }
  MODLE_DEV_MEMBER void load_blk(u32 t, Blk& r) const {
    const u32 w = word0(t);
    const u32 wq = w < n ? w : 0u;
    r.P = wave::ld4(pos, wq);
    r.M = wave::ld4(moves, wq);
    r.C = wave::ld4(coll, wq);
    r.B = wave::ld4(barpos, wq);
  }
  MODLE_DEV_MEMBER void init(Cell& c, BoundaryCounts bc, u32 list_cap, bool lef_bar, bool secondary) {
    Workspace& ws = c.ws;
    n = wave::uniform(c.n_active);
    lane = wave::lane();
    pos = FWD ? ws.f_pos : ws.r_pos;
    coll = FWD ? ws.f_coll : ws.r_coll;
    barpos = stalling_barrier_positions<FWD>(ws);
    moves = FWD ? ws.f_move : ws.r_move;
    // the candidate list (rank | flags) lives in a scratch array that is idle during the collision
    // passes; lanes with nothing to store hit a scratch word of their own (stores under a branch
    // cannot be counted by the compiler, and the wait for the next block's loads then becomes a
    // wait for every store in flight)
    q_k = FWD ? ws.tmp[1] : ws.tmp[0];
    dump = reinterpret_cast<u32*>(ws.sort_keys) + 2 * lane + (FWD ? 1 : 0);
    dump_m = reinterpret_cast<move_t*>(dump);
    cap = list_cap;
    correct_lef_bar = lef_bar;
    do_secondary = secondary;
    // rev: followers i = max(1, n5) .. n-1 ascending, blocker = rank i-1
    // fwd: followers i-1 for i = (n - min(n3, n3-1) - 1) .. 1 descending, blocker = rank i
    f_first = FWD ? static_cast<i32>(bc.n3 == 0 ? n - 1 : n - bc.n3) - 1
                  : static_cast<i32>(umax(1u, bc.n5));
    nblk = (n + 255) / 256;
    n_cand = 0;
    carry_pos = 0;
    carry_coll = 0;
    carry_pending = false;
    load_blk(0, cur);
  }
  // one block of 256 ranks
  MODLE_DEV_MEMBER void step(u32 t) {
    const Blk g = cur;
    if (t + 1 < nblk) load_blk(t + 1, cur);
    const u32 w = word0(t);
    u32 k[4], P[4], M[4], C[4];
    bool act[4];
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {  // j: position in visiting order inside the lane
      const u32 q = FWD ? 3 - j : j;
      k[j] = w + q;
      act[j] = k[j] < n;
      P[j] = act[j] ? g.P.v[q] : 0u;  // (ranks past the end: what follows the array; never used, and not passed around either)
      C[j] = act[j] ? g.C.v[q] : 0u;
      const u32 M0 = g.M.v[q];
      M[j] = M0;
      if (correct_lef_bar && act[j] && cw_occurred_as(C[j], EV_LEF_BAR)) {
        const u32 bp = g.B.v[q];
        M[j] = (FWD ? bp - P[j] : P[j] - bp) - 1;
      }
      *((act[j] && M[j] != M0) ? &moves[k[j]] : dump_m) = static_cast<move_t>(M[j]);
    }
    // the blocker of a unit: the unit visited before it
    const u32 pP_in = wave::shfl_up1(P[3]), pC_in = wave::shfl_up1(C[3]);
    bool pot[4], bocc[4];
    u32 fn = 2u;  // the lane's four steps composed (identity: p = 0, q = 1)
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      const u32 bP = j == 0 ? (lane > 0 ? pP_in : carry_pos) : P[j - 1];
      const u32 bC = j == 0 ? (lane > 0 ? pC_in : carry_coll) : C[j - 1];
      const i32 kk = static_cast<i32>(k[j]);
      const bool follower = do_secondary && act[j] && (FWD ? (kk <= f_first) : (kk >= f_first));
      pot[j] = follower && !cw_occurred(C[j]) &&
               (FWD ? static_cast<u64>(P[j]) + M[j] >= bP : static_cast<u64>(P[j]) - M[j] <= bP);
      bocc[j] = cw_occurred(bC);
      fn = sec_compose(fn, (pot[j] && bocc[j] ? 1u : 0u) | (pot[j] ? 2u : 0u));
    }
    // blocker stalled already, or itself a candidate (then it may become stalled in pass 2): the
    // unit before the block counts as "may be stalled" when it is a candidate (its outcome is not
    // known in this pass)
    const u32 incl = wave_prefix_sec_compose(fn);
    const u32 before_in = wave::shfl_up1(incl);
    const u32 before = lane > 0 ? before_in : 2u;  // the lanes before this one, composed
    bool x_prev = ((before & 1u) | ((before >> 1) & (carry_pending ? 1u : 0u))) != 0;
    bool x[4], cont[4];
    u32 lane_cnt = 0;
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      cont[j] = x_prev;
      x[j] = pot[j] && (bocc[j] || x_prev);
      x_prev = x[j];
      lane_cnt += x[j] ? 1u : 0u;
    }
    const u32 ps = wave_prefix_sum_u32(lane_cnt);
    u32 e = n_cand + ps - lane_cnt;
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      *((x[j] && e < cap) ? &q_k[e] : dump) = k[j] | (cont[j] ? SEC_CONT : 0u) | (bocc[j] ? SEC_BOCC : 0u);
      e += x[j] ? 1u : 0u;
    }
    n_cand += wave::bcast(ps, 63);
    carry_pending = wave::bcast(x[3], 63);
    carry_pos = wave::bcast(P[3], 63);
    carry_coll = wave::bcast(C[3], 63);
  }
};

// Pass 2 of one direction over the `n_cand` candidates pass 1 listed in ws.tmp[0] (rev) /
// ws.tmp[1] (fwd).
template <bool FWD>
MODLE_DEV_NOINLINE u32 secondary_resolve(Cell& c, u32 n_cand, u32* list, u32 list_cap, bool& overflow) {
  Workspace& ws = c.ws;
  const Params& p = *c.p;
  const u32 lane = wave::lane();
  const u32* pos = FWD ? ws.f_pos : ws.r_pos;
  const lefid_t* ids = FWD ? ws.f_id : ws.r_id;
  move_t* moves = FWD ? ws.f_move : ws.r_move;
  u32* coll = FWD ? ws.f_coll : ws.r_coll;
  const u32* const q_k = FWD ? ws.tmp[1] : ws.tmp[0];
  // run_lef_lef_collision_trial (simulation_impl.hpp:93-96): no draw when the bypass probability
  // is 0 (always collide) -- and none when it is 1: bernoulli_distribution(0) returns false
  // without touching the engine
  const f64 p_collide = 1.0 - p.p_bypass;
  const bool never_collide = p.p_bypass != 0.0 && p_collide == 0.0;
  const bool trials = p.p_bypass != 0.0 && !never_collide;
  if (n_cand == 0) return 0;
  if (n_cand > list_cap) {  // (cannot happen: a candidate is an active unit, the list holds capacity_lefs)
    c.error = ERR_INTERNAL;
    return 0;
  }

  u32 n_list = 0;
  u32 fin_pos = 0, fin_move = 0, fin_coll = 0, fin_id = 0;  // the candidate before this group, resolved
  // the units of a group and their blockers (the unit visited before: the adjacent rank) are
  // gathered through the list; the next group's list entries and units are requested one group
  // ahead.  (A blocker's move is final here: blockers that are candidates are taken from the lane
  // before, and everything else was settled by pass 1.)
  struct CandRegs {
    u32 K, P, M, C, I, bP, bM, bI;
  };
  const auto load_units_of = [&](u32 base, u32 kword, CandRegs& r) {
    const bool in = base + lane < n_cand;
    const u32 k = in ? (kword & CW_INDEX_MASK) : 0u;
    // (the first unit in visiting order is never a candidate: the adjacent rank exists)
    const u32 kb = in ? (FWD ? k + 1 : k - 1) : 0u;
    r.K = kword;
    r.P = wave::LdRaw{}(pos, k, true, 0, 0u);
    r.M = wave::LdRaw{}(moves, k, true, 0, 0u);
    r.C = wave::LdRaw{}(coll, k, true, 0, 0u);
    r.I = wave::LdRaw{}(ids, k, true, 0, 0u);
    r.bP = wave::LdRaw{}(pos, kb, true, 0, 0u);
    r.bM = wave::LdRaw{}(moves, kb, true, 0, 0u);
    r.bI = wave::LdRaw{}(ids, kb, true, 0, 0u);
  };
  const auto load_kword = [&](u32 base) { return wave::ld_sel(q_k, base + lane, base + lane < n_cand, 0u); };
  CandRegs ccur;
  load_units_of(0, load_kword(0), ccur);
  u32 kword_next = 64 < n_cand ? load_kword(64) : 0u;
  for (u32 base = 0; base < n_cand; base += 64) {
    const CandRegs q = ccur;
    if (base + 64 < n_cand) {
      load_units_of(base + 64, kword_next, ccur);
      if (base + 128 < n_cand) kword_next = load_kword(base + 128);
    }
    const u32 m = umin(64u, n_cand - base);
    const bool valid = lane < m;
    const u64 vmask = m == 64 ? ~u64(0) : lanemask_lt(m);
    const u32 k = q.K & CW_INDEX_MASK;
    const u32 P = q.P, id = q.I, M0 = q.M, C0 = q.C;
    u32 M = M0, C = C0;
    // blocker of the first lane when it is the last candidate of the previous group: resolved now
    bool cont = valid && (q.K & SEC_CONT) != 0;
    bool bocc = (q.K & SEC_BOCC) != 0;
    u32 xP = q.bP, xM = q.bM, xI = q.bI;  // explicit blocker (lanes that do not continue a chain)
    if (lane == 0 && cont) {
      xP = fin_pos;
      xM = fin_move;
      xI = fin_id;
      bocc = cw_occurred(fin_coll);
      cont = false;
    }
    const u64 contm = wave::ballot(cont);
    const u64 lt = lanemask_lt(lane), le = lt | (u64(1) << lane);
    const u64 starts = vmask & ~contm, ends = vmask & ~(contm >> 1);
    const u32 pP_in = wave::shfl_up1(P);
    const u32 blocker_pos = cont ? pP_in : xP;
    const auto wraps = [](u32 pp, u32 mm) { return FWD ? pp + mm < pp : mm > pp; };
    const bool odd = valid && (P == blocker_pos || wraps(P, M) || (!cont && wraps(xP, xM)));
    u64 pend = vmask;
    if (!wave::any(odd)) {
      const u32 bI_in = wave::shfl_up1(id);
      const u32 bId = cont ? bI_in : xI;
      const u32 s = valid ? static_cast<u32>(63 - wave::clz64(starts & le)) : lane;  // start of the lane's run
      // landing position and state of the run's own blocker (explicit at the run's first lane)
      const u32 xland = FWD ? xP + xM : xP - xM;
      const u32 lb = wave::shfl(xland, s);
      const bool head_ok = wave::shfl(static_cast<u32>(bocc), s) != 0;
      const u32 off = lane - s;
      const u32 land_prev = FWD ? lb - off : lb + off;  // the blocker's landing while the chain holds
      const bool geo = FWD ? (P + M >= land_prev) : (P - M <= land_prev);
      // a run whose first blocker is not stalled does nothing at all
      const u64 ngeo = wave::ballot(valid && (!geo || (lane == s && !head_ok)));
      const bool alive = valid && ((ngeo & le) >> s) == 0;
      // the lanes that draw unless an "avoid" before them ends their chain: in every run a
      // prefix of its lanes
      const u64 live = wave::ballot(alive);
      // outcome of stream output t, for the first 64 outputs (at most popc(live) are consumed)
      u64 outcomes = never_collide ? u64(0) : ~u64(0);
      if (trials && live != 0) {
        rng_ensure(c.g, static_cast<u32>(wave::popc64(live)));
        outcomes = wave::ballot(bernoulli_raw(rng_peek(c.g, c.g.pos + lane), p_collide));
      }
      u64 hits = 0, avoids = 0;  // lanes that collide / whose collision is avoided
      u32 drawn = 0;             // outputs consumed
      // Every lane takes the output its position among the drawing lanes gives it.  That is final
      // up to the first "avoid" that ends a chain with lanes still to draw behind it (those lanes
      // drop out, and every later lane moves to an earlier output): one round per such avoid, and
      // most avoids are the last lane of their chain.
      const u64 has_successor = (live >> 1) & (contm >> 1);  // the next lane draws after this one, same chain
      for (u64 rem = live; rem != 0;) {
        const u32 t = drawn + static_cast<u32>(wave::popc64(rem & lt));
        const bool collide = ((outcomes >> (t & 63u)) & 1u) != 0;
        const u64 av = wave::ballot(((rem >> lane) & 1u) != 0 && !collide);
        const u64 cut = av & has_successor;
        if (cut == 0) {
          hits |= rem & ~av;
          avoids |= av;
          drawn += static_cast<u32>(wave::popc64(rem));
          break;
        }
        const u32 a = static_cast<u32>(wave::ctz64(cut));
        const u32 e = static_cast<u32>(wave::ctz64(ends & ~lanemask_lt(a)));  // end of its run
        const u64 upto = lanemask_lt(a) | (u64(1) << a);
        hits |= rem & upto & ~av;
        avoids |= av & upto;
        drawn += static_cast<u32>(wave::popc64(rem & upto));
        rem = e >= 63 ? u64(0) : rem & ~lanemask_lt(e + 1);
      }
      if (trials && drawn != 0) rng_advance(c.g, drawn);
      if ((avoids >> lane) & 1u) {
        C = cw_make(bId, EV_LEF_LEF_SECONDARY);
        const u32 j = n_list + static_cast<u32>(wave::popc64(avoids & lt));
        if (j < list_cap) list[j] = k;
        if (c.filter_on) {
          rank_filter_add_id(c, id);
          rank_filter_add_id(c, bId);
        }
      }
      n_list += static_cast<u32>(wave::popc64(avoids));
      if (n_list > list_cap) overflow = true;
      if ((hits >> lane) & 1u) {
        const u32 move = FWD ? land_prev - P : P - land_prev;
        M = umin(move, move - 1);
        C = cw_make(bId, EV_COLLISION | EV_LEF_LEF_SECONDARY);
      }
      pend = 0;
    }
    // Rounds (a candidate AT its blocker's position, moves that wrap): a lane is ready when its
    // blocker -- the lane before it for a chain lane -- is resolved; all ready lanes below the first
    // lane that still waits are resolved together, their draws numbered in lane order.
    while (pend != 0) {
      const u32 pP = wave::shfl_up1(P), pM = wave::shfl_up1(M);
      const u32 pC = wave::shfl_up1(C), pI = wave::shfl_up1(id);
      const u32 bP = cont ? pP : xP, bM = cont ? pM : xM, bId = cont ? pI : xI;
      const bool b_stalled = cont ? cw_occurred(pC) : bocc;
      const u64 ready = pend & ~((pend << 1) & contm);
      const u64 waiting = pend & ~ready;
      const u64 now =
          waiting != 0 ? (ready & lanemask_lt(static_cast<u32>(wave::ctz64(waiting)))) : ready;
      const bool mine = ((now >> lane) & 1u) != 0;
      const bool geo = FWD ? (static_cast<u64>(P) + M >= static_cast<u64>(bP) + bM)
                           : (static_cast<u64>(P) - M <= static_cast<u64>(bP) - bM);
      const bool draws = mine && b_stalled && geo;
      const u64 dm = wave::ballot(draws);
      bool collide = draws && !never_collide;
      if (trials && dm != 0) {
        const u32 cnt = static_cast<u32>(wave::popc64(dm));
        rng_ensure(c.g, cnt);
        const u32 t = static_cast<u32>(wave::popc64(dm & lanemask_lt(lane)));
        collide = draws && bernoulli_raw(rng_peek(c.g, c.g.pos + t), p_collide);
        rng_advance(c.g, cnt);
      }
      const bool avoided = draws && !collide;
      if (collide) {
        const u32 move = FWD ? (bP + bM) - P : P - (bP - bM);
        M = umin(move, move - 1);
        C = cw_make(bId, EV_COLLISION | EV_LEF_LEF_SECONDARY);
      }
      const u64 am = wave::ballot(avoided);
      if (avoided) {
        C = cw_make(bId, EV_LEF_LEF_SECONDARY);
        const u32 j = n_list + static_cast<u32>(wave::popc64(am & lanemask_lt(lane)));
        if (j < list_cap) list[j] = k;
        if (c.filter_on) {
          rank_filter_add_id(c, id);
          rank_filter_add_id(c, bId);
        }
      }
      n_list += static_cast<u32>(wave::popc64(am));
      if (n_list > list_cap) overflow = true;
      pend &= ~now;
    }
    if (valid && (M != M0 || C != C0)) {
      moves[k] = M;
      coll[k] = C;
    }
    fin_pos = wave::bcast(P, m - 1);
    fin_move = wave::bcast(M, m - 1);
    fin_coll = wave::bcast(C, m - 1);
    fin_id = wave::bcast(id, m - 1);
  }
  wave::sync_mem();
  return n_list;
}

// one direction: filter, then resolve (phase-level hooks; the epoch loop runs the two filters in one loop)
template <bool FWD>
MODLE_DEV_NOINLINE u32 process_secondary(Cell& c, BoundaryCounts bc, u32* list, u32 list_cap,
                                         bool& overflow, bool correct_lef_bar, bool do_secondary) {
  SecondaryFilter<FWD> f;
  f.init(c, bc, list_cap, correct_lef_bar, do_secondary);
  for (u32 t = 0; t < f.nblk; ++t) f.step(t);
  wave::sync_mem();
  return secondary_resolve<FWD>(c, f.n_cand, list, list_cap, overflow);
}

// both directions: the two filters in one loop, then the rev and the fwd resolve pass (draw order)
MODLE_DEV_NOINLINE void process_secondary_both(Cell& c, BoundaryCounts bc, u32* list_rev, u32* list_fwd,
                                               u32 list_cap, bool& overflow, u32& n_rev, u32& n_fwd) {
  // helper-wave mode (sim_pair.h): the fwd filter runs on the helper during the rev filter and
  // the rev resolve pass (the resolve passes draw: rev first, then fwd, both on this wave)
  SecondaryFilter<false> fr;
  fr.init(c, bc, list_cap, true, true);
  u32 n_cand_fwd = 0;
  if (c.pair_on) {
    pair_request_sec_filter(c, bc.n5, bc.n3, list_cap);
    for (u32 t = 0; t < fr.nblk; ++t) fr.step(t);
  } else {
    SecondaryFilter<true> ff;
    ff.init(c, bc, list_cap, true, true);
    for (u32 t = 0; t < fr.nblk; ++t) {
      fr.step(t);
      ff.step(t);
    }
    n_cand_fwd = ff.n_cand;
  }
  wave::sync_mem();
  n_rev = secondary_resolve<false>(c, fr.n_cand, list_rev, list_cap, overflow);
  if (c.pair_on) n_cand_fwd = pair_take_sec_filter(c);
  n_fwd = secondary_resolve<true>(c, n_cand_fwd, list_fwd, list_cap, overflow);
}

// fix_secondary_lef_lef_collisions (reference: simulation_detect_collisions.cpp:517-644).
// Rare (one entry per avoided secondary collision); replayed sequentially, uniformly.  The two
// units trade places: slots i-1 and i of the rank-ordered arrays are rewritten.
MODLE_DEV_NOINLINE void fix_secondary_rev_seq(Cell& c, const u32* list, u32 n_list) {
  Workspace& ws = c.ws;
  const u32 start = c.iv->start;
  const u32 sec = EV_LEF_LEF_SECONDARY;
  for (u32 q = 0; q < n_list; ++q) {  // list is in ascending rank order
    const u32 i = list[q];
    if (!cw_avoided_as(ws.r_coll[i], sec)) continue;
    const u32 id1 = ws.r_id[i - 1], id2 = ws.r_id[i];
    const u32 p1 = ws.r_pos[i - 1], p2 = ws.r_pos[i];
    const u32 m1 = ws.r_move[i - 1];
    const u32 c1 = ws.r_coll[i - 1];
    const u32 pos1 = p1 - m1;
    const u32 m2 = p2 > pos1 + 1 ? p2 - (pos1 + 1) : 0;
    const u32 c2 = cw_make(id1, EV_COLLISION | sec);
    const u32 np1 = umin(c.by_id_valid ? ws.by_id_pos[1][id1] : ws.f_pos[ws.f_rank[id1]], p2);
    const u32 np2 = umin(c.by_id_valid ? ws.by_id_pos[1][id2] : ws.f_pos[ws.f_rank[id2]], p1);
    wave::lockstep();
    // unit 2 moves to slot i-1 with unit 1's old collision / move, unit 1 to slot i
    ws.r_id[i - 1] = id2;
    ws.r_pos[i - 1] = np2;
    ws.r_coll[i - 1] = c1;
    ws.r_move[i - 1] = umin(np2 - start, m1);
    ws.r_id[i] = id1;
    ws.r_pos[i] = np1;
    ws.r_coll[i] = c2;
    ws.r_move[i] = umin(np1 - start, m2);
    ws.r_rank[id2] = i - 1;
    ws.r_rank[id1] = i;
    if (c.by_id_valid) {
      ws.by_id_pos[0][id2] = np2;
      ws.by_id_pos[0][id1] = np1;
    }
    wave::sync_mem();
  }
}

MODLE_DEV_NOINLINE void fix_secondary_fwd_seq(Cell& c, const u32* list, u32 n_list) {
  Workspace& ws = c.ws;
  const u32 last = c.iv->end - 1;
  const u32 sec = EV_LEF_LEF_SECONDARY;
  for (u32 q = n_list; q-- > 0;) {  // list is in descending rank order; the fix loop ascends
    const u32 i = list[q];
    if (!cw_avoided_as(ws.f_coll[i], sec)) continue;
    const u32 id1 = ws.f_id[i], id2 = ws.f_id[i + 1];
    const u32 p1 = ws.f_pos[i], p2 = ws.f_pos[i + 1];
    const u32 m2 = ws.f_move[i + 1];
    const u32 c2 = ws.f_coll[i + 1];
    const u32 pos2 = p2 + m2;
    const u32 m1 = pos2 > p1 + 1 ? pos2 - (p1 + 1) : 0;
    const u32 c1 = cw_make(id2, EV_COLLISION | sec);
    const u32 np1 = umax(c.by_id_valid ? ws.by_id_pos[0][id1] : ws.r_pos[ws.r_rank[id1]], p2);
    const u32 np2 = umax(c.by_id_valid ? ws.by_id_pos[0][id2] : ws.r_pos[ws.r_rank[id2]], p1);
    wave::lockstep();
    ws.f_id[i] = id2;
    ws.f_pos[i] = np2;
    ws.f_coll[i] = c1;
    ws.f_move[i] = umin(last - np2, m1);
    ws.f_id[i + 1] = id1;
    ws.f_pos[i + 1] = np1;
    ws.f_coll[i + 1] = c2;
    ws.f_move[i + 1] = umin(last - np1, m2);
    ws.f_rank[id2] = i;
    ws.f_rank[id1] = i + 1;
    if (c.by_id_valid) {
      ws.by_id_pos[1][id2] = np2;
      ws.by_id_pos[1][id1] = np1;
    }
    wave::sync_mem();
  }
}

// Entries of the list touch the rank slots {i-1, i} (rev) / {i, i+1} (fwd).  Unless two entries are
// adjacent ranks the swaps are independent of each other and every lane performs one; a list
// with adjacent entries (a cascade of avoided collisions) is replayed sequentially.
MODLE_DEV bool fix_list_has_adjacent_entries(const u32* list, u32 n_list, bool ascending) {
  const u32 lane = wave::lane();
  bool adj = false;
  for (u32 base = 0; base < n_list; base += 64) {
    const u32 q = base + lane;
    bool a = false;
    if (q + 1 < n_list) {
      const u32 x = list[q], y = list[q + 1];
      a = ascending ? (y <= x + 1) : (x <= y + 1);
    }
    adj = wave::any(a) || adj;
  }
  return adj;
}

MODLE_DEV_NOINLINE void fix_secondary_rev(Cell& c, const u32* list, u32 n_list) {
  if (fix_list_has_adjacent_entries(list, n_list, true)) {
    fix_secondary_rev_seq(c, list, n_list);
    return;
  }
  Workspace& ws = c.ws;
  const u32 lane = wave::lane();
  const u32 start = c.iv->start;
  const u32 sec = EV_LEF_LEF_SECONDARY;
  for (u32 base = 0; base < n_list; base += 64) {
    const u32 q = base + lane;
    if (q < n_list) {
      const u32 i = list[q];
      if (cw_avoided_as(ws.r_coll[i], sec)) {
        const u32 id1 = ws.r_id[i - 1], id2 = ws.r_id[i];
        const u32 p1 = ws.r_pos[i - 1], p2 = ws.r_pos[i];
        const u32 m1 = ws.r_move[i - 1];
        const u32 c1 = ws.r_coll[i - 1];
        const u32 pos1 = p1 - m1;
        const u32 m2 = p2 > pos1 + 1 ? p2 - (pos1 + 1) : 0;
        const u32 c2 = cw_make(id1, EV_COLLISION | sec);
        const u32 np1 = umin(c.by_id_valid ? ws.by_id_pos[1][id1] : ws.f_pos[ws.f_rank[id1]], p2);
        const u32 np2 = umin(c.by_id_valid ? ws.by_id_pos[1][id2] : ws.f_pos[ws.f_rank[id2]], p1);
        ws.r_id[i - 1] = id2;
        ws.r_pos[i - 1] = np2;
        ws.r_coll[i - 1] = c1;
        ws.r_move[i - 1] = umin(np2 - start, m1);
        ws.r_id[i] = id1;
        ws.r_pos[i] = np1;
        ws.r_coll[i] = c2;
        ws.r_move[i] = umin(np1 - start, m2);
        ws.r_rank[id2] = i - 1;
        ws.r_rank[id1] = i;
        if (c.by_id_valid) {
          ws.by_id_pos[0][id2] = np2;
          ws.by_id_pos[0][id1] = np1;
        }
      }
    }
  }
  wave::sync_mem();
}

MODLE_DEV_NOINLINE void fix_secondary_fwd(Cell& c, const u32* list, u32 n_list) {
  if (fix_list_has_adjacent_entries(list, n_list, false)) {
    fix_secondary_fwd_seq(c, list, n_list);
    return;
  }
  Workspace& ws = c.ws;
  const u32 lane = wave::lane();
  const u32 last = c.iv->end - 1;
  const u32 sec = EV_LEF_LEF_SECONDARY;
  for (u32 base = 0; base < n_list; base += 64) {
    const u32 q = base + lane;
    if (q < n_list) {
      const u32 i = list[q];
      if (cw_avoided_as(ws.f_coll[i], sec)) {
        const u32 id1 = ws.f_id[i], id2 = ws.f_id[i + 1];
        const u32 p1 = ws.f_pos[i], p2 = ws.f_pos[i + 1];
        const u32 m2 = ws.f_move[i + 1];
        const u32 c2 = ws.f_coll[i + 1];
        const u32 pos2 = p2 + m2;
        const u32 m1 = pos2 > p1 + 1 ? pos2 - (p1 + 1) : 0;
        const u32 c1 = cw_make(id2, EV_COLLISION | sec);
        const u32 np1 = umax(c.by_id_valid ? ws.by_id_pos[0][id1] : ws.r_pos[ws.r_rank[id1]], p2);
        const u32 np2 = umax(c.by_id_valid ? ws.by_id_pos[0][id2] : ws.r_pos[ws.r_rank[id2]], p1);
        ws.f_id[i] = id2;
        ws.f_pos[i] = np2;
        ws.f_coll[i] = c1;
        ws.f_move[i] = umin(last - np2, m1);
        ws.f_id[i + 1] = id1;
        ws.f_pos[i + 1] = np1;
        ws.f_coll[i + 1] = c2;
        ws.f_move[i + 1] = umin(last - np1, m2);
        ws.f_rank[id2] = i;
        ws.f_rank[id1] = i + 1;
        if (c.by_id_valid) {
          ws.by_id_pos[1][id2] = np2;
          ws.by_id_pos[1][id1] = np1;
        }
      }
    }
  }
  wave::sync_mem();
}

// fix_secondary needs the OTHER unit of the two LEFs of every list entry: entries of the rev list
// the fwd units, entries of the fwd list the rev units.  Without a complete inverse permutation
// their ranks come from ONE sweep over both id arrays (four ranks per lane, four blocks of loads in
// flight per direction) against the bitmap of LEF ids the secondary pass has collected in LDS:
// ws.r_rank / ws.f_rank then hold valid entries for those LEFs.  (The rev fix re-orders rev units
// before the fwd fix looks at them, but it updates ws.r_rank for every unit it moves; the ids on
// the slots a cascade of fixes touches are the ids of its entries, whatever their order.)
MODLE_DEV_NOINLINE void lookup_partner_ranks(Cell& c, bool want_r, bool want_f) {
  // (the positions by LEF id are at hand: nothing to look up)
  want_r = want_r && !c.inv_valid[0] && !c.by_id_valid;
  want_f = want_f && !c.inv_valid[1] && !c.by_id_valid;
  if (!want_f && !want_r) return;
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  wave::sync_lds();
  constexpr u32 GB = 4;  // blocks of 256 ranks per group of loads
  const u32 nblk = (n + 255) / 256;
  for (u32 t0 = 0; t0 < nblk; t0 += GB) {
    wave::U32x4 R[GB], F[GB];
#pragma unroll
    for (u32 g = 0; g < GB; ++g) {
      const u32 w = 256 * (t0 + g) + 4 * lane;
      R[g] = wave::ld4(ws.r_id, (want_r && w < n) ? w : 0u);
      F[g] = wave::ld4(ws.f_id, (want_f && w < n) ? w : 0u);
    }
#pragma unroll
    for (u32 g = 0; g < GB; ++g) {
      const u32 w = 256 * (t0 + g) + 4 * lane;
#pragma unroll
      for (u32 q = 0; q < 4; ++q) {
        const bool in = w + q < n;
        if (want_r && in && rank_filter_test(c, R[g].v[q])) ws.r_rank[R[g].v[q]] = w + q;
        if (want_f && in && rank_filter_test(c, F[g].v[q])) ws.f_rank[F[g].v[q]] = w + q;
      }
    }
  }
  wave::sync_mem();
}

// returns false when an internal capacity was exceeded (the cell is then flagged as failed)
MODLE_DEV bool phase_process_collisions(Cell& c) {
  BoundaryCounts bc;
  PHASE(c, 8, bc = detect_boundaries(c));
  // helper-wave mode (sim_pair.h): without Bernoulli trials the two instances draw nothing and touch
  // disjoint arrays; the helper takes the fwd one
  const bool split = c.pair_on && !lef_bar_trials_needed(*c.p) && wave::uniform(c.iv->n_barriers) != 0;
  if (split) {
    bool handed = false;
    PHASE(c, 9, pair_request_lef_bar(c, bc.n5, bc.n3); detect_lef_bar<false>(c, bc); handed = pair_wait(c, PAIR_ALL));
    if (!handed) return false;  // (c.error says why: sim_pair.h)
  } else {
    PHASE(c, 9, detect_lef_bar<false>(c, bc); detect_lef_bar<true>(c, bc));
  }
  PHASE(c, 10, detect_primary(c, bc, true));
  bool overflow = false;
  // avoided secondary collisions are listed in device scratch: one entry per unit at most
  u32* list_rev = c.ws.tmp[5];
  u32* list_fwd = c.ws.tmp[6];
  const u32 cap = c.ws.capacity_lefs;
  u32 nr = 0, nf = 0;
  // (the LDS sort buffer is idle from here to the release: it holds the id filter)
  c.filter_on = !(c.inv_valid[0] && c.inv_valid[1]);
  if (c.filter_on) rank_filter_clear(c, c.n_active);
  PHASE(c, 11, process_secondary_both(c, bc, list_rev, list_fwd, cap, overflow, nr, nf));
  c.filter_on = false;
  if (overflow) c.error = ERR_LIST_OVERFLOW;
  if (c.error != 0) return false;
  PHASE(c, 12, if ((nr | nf) != 0) lookup_partner_ranks(c, nf != 0, nr != 0);
        if (nr != 0) fix_secondary_rev(c, list_rev, nr);
        if (nf != 0) fix_secondary_fwd(c, list_fwd, nf));
  return true;
}

}  // namespace modle_dev

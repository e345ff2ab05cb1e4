// sim_bind_rank.h -- part of sim_device.h (included by it, in this order): select_and_bind_lefs and rank_lefs.
#pragma once

namespace modle_dev {

// =============================================================================================
// select_and_bind_lefs (reference: simulation.cpp:988-993, simulation_impl.hpp:30-91)
// =============================================================================================
MODLE_DEV_NOINLINE void phase_bind(Cell& c, u32 epoch_now) {
  const Interval& iv = *c.iv;
  c.keys_valid = false;
  ensure_inverse_both(c);
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  const u64 range = static_cast<u64>(iv.end) - 1 - iv.start;
  const u64 bucket = range != 0 ? uniform_int_bucket(range) : 1;
  // bucket >= 2^32 here (range < 2^32), so quotients stay below 2^32 + 1: see udiv_by_uniform
  const bool fast_div = bucket <= (u64(1) << 62) && bucket >= (u64(1) << 24);
  const f64 inv_bucket = 1.0 / static_cast<f64>(bucket);
  constexpr u32 UX = 4;  // batches per group; the next group's loads go before this group's stores
  struct LefRegs {
    u32 E[UX], R[UX], F[UX];
  };
  const auto load_lefs = [&](auto op, u32 group, LefRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 iq = group + 64 * u + lane;
      r.E[u] = op(ws.epoch, iq, iq < n, 0, r.E[u]);
      r.R[u] = op(ws.r_rank, iq, iq < n, 0, r.R[u]);
      r.F[u] = op(ws.f_rank, iq, iq < n, 0, r.F[u]);
    }
  };
  LefRegs cur;
  load_lefs(wave::LdRaw{}, 0, cur);
  for (u32 group = 0; group < n; group += 64 * UX) {
    LefRegs g = cur;
    load_lefs(wave::LdMask{}, group, g);  // (defaults of the lanes outside the range)
    if (group + 64 * UX < n) load_lefs(wave::LdRaw{}, group + 64 * UX, cur);
    const u32* Eq = g.E;
    const u32* Rq = g.R;
    const u32* Fq = g.F;
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
    const u32 base = group + 64 * u;
    if (base >= n) break;
    const u32 i = base + lane;
    const bool unb = i < n && Eq[u] == UNBOUND;
    const u64 mask = wave::ballot(unb);
    if (mask == 0) continue;
    u32 posv = iv.start;
    if (range != 0) {
      const u32 cnt = static_cast<u32>(wave::popc64(mask));
      rng_ensure(c.g, cnt);
      const u32 k = static_cast<u32>(wave::popc64(mask & lanemask_lt(lane)));
      const u64 raw = rng_peek(c.g, c.g.pos + k);
      const u64 r = fast_div ? udiv_by_uniform(raw, bucket, inv_bucket) : raw / bucket;
      if (wave::any(unb && r > range)) {
        // a draw was rejected (p ~ range / 2^64): replay the batch sequentially
        u64 m = mask;
        while (m != 0) {
          const u32 l = static_cast<u32>(wave::ctz64(m));
          m &= m - 1;
          const u64 v = uniform_int_exact(c.g, range, bucket);
          if (lane == l) posv = iv.start + static_cast<u32>(v);
        }
      } else {
        posv = iv.start + static_cast<u32>(r);
        rng_advance(c.g, cnt);
      }
    }
    if (unb) {
      ws.epoch[i] = epoch_now;
      const u32 kr = Rq[u], kf = Fq[u];
      ws.r_pos[kr] = posv;
      ws.r_move[kr] = NEW_MARK;
      ws.f_pos[kf] = posv;
      ws.f_move[kf] = NEW_MARK;
    }
    }
  }
  wave::sync_mem();
}

// The same from the list release_lefs left in LDS: inside the epoch loop the LEFs to bind are
// exactly the ones released in the previous epoch (ascending ids) followed by the ones activated
// since the last bind (ids n_bound .. n_active-1, never ranked: their slots are the identity).
// No sweep over the LEFs; the ranks of the listed LEFs are the only thing read.
MODLE_DEV_NOINLINE void phase_bind_listed(Cell& c, u32 epoch_now) {
  const Interval& iv = *c.iv;
  Workspace& ws = c.ws;
  const u32 lane = wave::lane();
  const u32 n_rel = wave::uniform(c.n_rel);
  const u32 first_new = wave::uniform(c.n_bound);
  const u32 total = n_rel + (wave::uniform(c.n_active) - first_new);
  const u64 range = static_cast<u64>(iv.end) - 1 - iv.start;
  const u64 bucket = range != 0 ? uniform_int_bucket(range) : 1;
  const bool fast_div = bucket <= (u64(1) << 62) && bucket >= (u64(1) << 24);
  const f64 inv_bucket = 1.0 / static_cast<f64>(bucket);
  const u32* list = reinterpret_cast<const u32*>(c.lds.sort_lds);
  u64* keys_rev = reinterpret_cast<u64*>(ws.tmp[2]);  // (two arrays each: capacity >= total keys)
  u64* keys_fwd = reinterpret_cast<u64*>(ws.tmp[4]);
  for (u32 base = 0; base < total; base += 64) {
    const u32 e = base + lane;
    const bool act = e < total;
    const bool listed = e < n_rel;
    const u32 id = listed ? list[e] : first_new + (e - n_rel);
    u32 kr = id, kf = id;
    if (act && listed) {
      kr = ws.r_rank[id];
      kf = ws.f_rank[id];
    }
    u32 posv = iv.start;
    if (range != 0) {
      const u32 cnt = umin(64u, total - base);
      rng_ensure(c.g, cnt);
      const u64 raw = rng_peek(c.g, c.g.pos + lane);
      const u64 r = fast_div ? udiv_by_uniform(raw, bucket, inv_bucket) : raw / bucket;
      if (wave::any(act && r > range)) {
        // a draw was rejected (p ~ range / 2^64): replay the batch sequentially
        for (u32 l = 0; l < cnt; ++l) {
          const u64 v = uniform_int_exact(c.g, range, bucket);
          if (lane == l) posv = iv.start + static_cast<u32>(v);
        }
      } else {
        posv = iv.start + static_cast<u32>(r);
        rng_advance(c.g, cnt);
      }
    }
    if (act) {
      ws.epoch[id] = epoch_now;
      ws.r_pos[kr] = posv;
      ws.r_move[kr] = NEW_MARK;
      ws.f_pos[kf] = posv;
      ws.f_move[kf] = NEW_MARK;
      keys_rev[e] = (static_cast<u64>(posv) << 32) | kr;
      keys_fwd[e] = (static_cast<u64>(posv) << 32) | kf;
    }
  }
  c.n_rel = 0;
  c.n_bound = c.n_active;
  c.n_keys = total;
  c.keys_valid = true;
  wave::sync_mem();
}

// =============================================================================================
// rank_lefs (reference: simulation.cpp:410-496)
//
// Total order: position, then binding epoch (rev: older first, fwd: younger first), then the
// position in the incoming rank order (the reference leaves this last tie to an unstable sort;
// DESIGN.md "ranking ties").  Units that were already ranked stay sorted across an epoch except
// where fix_secondary_lef_lef_collisions re-positions a pair, so the update is: split the rank
// order into carried-over units that are still in order and "new" units (bound this epoch, or
// out of order), sort the new ones, merge, then order equal positions.
// =============================================================================================
MODLE_DEV u32 pow2_ceil(u32 x) {
  u32 p = 1;
  while (p < x) p <<= 1;
  return p;
}

template <bool IN_LDS>
MODLE_DEV_NOINLINE void bitonic_sort_u64(u64* keys, u32 m_pow2) {
  const u32 lane = wave::lane();
  const u32 half = m_pow2 / 2;
  for (u32 k = 2; k <= m_pow2; k <<= 1) {
    for (u32 j = k >> 1; j > 0; j >>= 1) {
      for (u32 base = 0; base < half; base += 64) {
        const u32 t = base + lane;
        if (t < half) {
          const u32 i = (t / j) * 2 * j + (t % j);
          const u32 l = i + j;
          const bool up = (i & k) == 0;
          const u64 a = keys[i], b = keys[l];
          if ((a > b) == up) {
            keys[i] = b;
            keys[l] = a;
          }
        }
      }
      if (IN_LDS) wave::sync_lds(); else wave::sync_mem();
    }
  }
}

// full comparator on (pos, id) pairs: position, binding epoch (rev: older first, fwd: younger
// first), previous rank (`where`, by LEF id)
template <bool FWD>
MODLE_DEV bool rank_pair_out_of_order(const Workspace& ws, const u32* where, u32 pa, u32 ida,
                                      u32 pb, u32 idb) {
  if (pa != pb) return pa > pb;
  const u32 ea = ws.epoch[ida], eb = ws.epoch[idb];
  if (ea != eb) return FWD ? ea < eb : ea > eb;
  return where[ida] > where[idb];
}

// Merge step of rank_update: kept units (old_pos / old_id, sorted) and the sorted keys of the new
// units go to their final ranks; returns true when two bound units share a position.  The
// loads of the next batch are issued before the (scattered) stores of the current one: on
// this hardware a wait for a load also waits for every store issued before it.
template <bool FWD>
MODLE_DEV bool rank_merge(const u64* keys, u32 n_new, u32 n_old, const u32* old_pos,
                          const lefid_t* old_id, const lefid_t* new_id, u32* out_pos, lefid_t* out_id,
                          u32* where_new, u32* cnt_lds) {
  const u32 lane = wave::lane();
  bool ties = false;
  // cnt_lds[j] = number of kept units that go before new key j, filled in while the kept units
  // are placed (they see where the keys fall between them); keys after the last kept unit keep
  // the initial value.  Only when the keys fit the buffer; otherwise the keys search old_pos.
  const bool use_cnt = n_new <= STAGE_CAP;
  if (use_cnt) {
    wave::lockstep();
    for (u32 j = lane; j < n_new; j += 64) cnt_lds[j] = n_old;
    wave::sync_lds();
  }
  u32 carry_lo = 0;  // keys below the last kept unit of the previous batch

  u32 carry_old = UNBOUND;  // position of the kept unit before this batch (UNBOUND: none)
  constexpr u32 UX = 4;  // batches per group
  struct KeptRegs {
    u32 P[UX], I[UX];
  };
  const auto load_kept = [&](auto op, u32 group, KeptRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 aq = group + 64 * u + lane;
      r.P[u] = op(old_pos, aq, aq < n_old, UNBOUND, r.P[u]);
      r.I[u] = op(old_id, aq, aq < n_old, 0, r.I[u]);
    }
  };
  KeptRegs cur;
  load_kept(wave::LdRaw{}, 0, cur);
  for (u32 group = 0; group < n_old; group += 64 * UX) {
    KeptRegs g = cur;
    load_kept(wave::LdMask{}, group, g);  // (defaults of the lanes outside the range)
    if (group + 64 * UX < n_old) load_kept(wave::LdRaw{}, group + 64 * UX, cur);
    const u32* Pq = g.P;
    const u32* Iq = g.I;
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
    const u32 base = group + 64 * u;
    if (base >= n_old) break;
    const u32 a = base + lane;
    const bool act = a < n_old;
    const u32 pp = Pq[u];
    const u32 oid = Iq[u];
    bool tie = false;
    // lo = number of keys that go before this unit.  Kept units and keys are both sorted, so the
    // search continues from the previous batch's last answer: a few fixed steps reach almost
    // every unit (a batch of 64 kept units has a couple of keys between them), the rest finish
    // with a binary search
    u32 lo = act ? carry_lo : 0;
    if (act) {
      const u64 thr = FWD ? ((static_cast<u64>(pp) + 1) << 32) : (static_cast<u64>(pp) << 32);
#pragma unroll
      for (u32 sft = 8; sft >= 1; sft >>= 1) {
        const u32 j = lo + sft;
        const bool in = j <= n_new;
        const u64 kv = keys[in ? j - 1 : 0];  // (no branch around the read)
        if (in & (kv < thr)) lo = j;
      }
      if (lo == carry_lo + 15 && lo < n_new) {
        u32 hi = n_new;
        while (lo < hi) {
          const u32 mid = (lo + hi) >> 1;
          if (keys[mid] < thr) lo = mid + 1; else hi = mid;
        }
      }
    }
    const u32 lo_first = carry_lo;
    {
      const u64 am = wave::ballot(act);
      carry_lo = wave::bcast(lo, static_cast<u32>(63 - wave::clz64(am)));
    }
    if (use_cnt) {
      // keys [lo of the previous kept unit, lo) lie between that unit and this one
      const u32 lo_in = wave::shfl_up1(lo);
      const u32 lo_prev = lane > 0 ? lo_in : lo_first;
      if (act) {
        for (u32 j = lo_prev; j < lo; ++j) cnt_lds[j] = a;
      }
    }
    if (act) {
      if (pp != UNBOUND) {
        if (FWD) {
          tie = lo > 0 && static_cast<u32>(keys[lo - 1] >> 32) == pp;
        } else {
          tie = lo < n_new && static_cast<u32>(keys[lo] >> 32) == pp;
        }
      }
      wave::st_stream(&out_pos[a + lo], pp);
      wave::st_stream(&out_id[a + lo], oid);
      where_new[oid] = a + lo;
    }
    const u32 prev_in = wave::shfl_up1(pp);
    const u32 prev = lane > 0 ? prev_in : carry_old;
    tie = tie || (act && pp != UNBOUND && prev == pp && (base != 0 || lane != 0));
    ties = wave::any(tie) || ties;
    carry_old = wave::bcast(pp, 63);
    }
  }
  wave::sync_lds();
  for (u32 base = 0; base < n_new; base += 64) {
    const u32 bq = base + lane;
    bool tie = false;
    if (bq < n_new) {
      const u64 key = keys[bq];
      const u32 pp = static_cast<u32>(key >> 32);
      u32 lo = 0;
      if (use_cnt) {
        lo = cnt_lds[bq];
      } else {
        u32 hi = n_old;
        while (lo < hi) {
          const u32 mid = (lo + hi) >> 1;
          const u32 q = old_pos[mid];
          const bool before = FWD ? (q < pp) : (q <= pp);
          if (before) lo = mid + 1; else hi = mid;
        }
      }
      const u32 nid = new_id[static_cast<u32>(key)];
      wave::st_stream(&out_pos[bq + lo], pp);
      wave::st_stream(&out_id[bq + lo], nid);
      where_new[nid] = bq + lo;
      tie = bq + 1 < n_new && static_cast<u32>(keys[bq + 1] >> 32) == pp;
    }
    ties = wave::any(tie) || ties;
  }
  return ties;
}

// Steps 4 and 5 of a rank update: order equal positions, make the new arrays current.
// Every pair of neighbours with equal positions lies inside the output slots [t_lo, t_hi] (the
// sweeps flag at least one member of every such pair): the transposition passes stay inside that
// range (one slot of margin on both sides).
// `where` != nullptr (general update): the previous ranks by LEF id are the last tie-break and the
// new inverse permutation (ws.tmp[7]) is kept up to date and made current.
// `where` == nullptr (update of the epoch loop): the merge has left equal positions in the order
// of their previous ranks, so a STABLE ordering by binding epoch is the full comparator; no
// inverse permutation is written.
template <bool FWD>
MODLE_DEV void rank_finish(Cell& c, bool ties, const u32* where, u32 t_lo, u32 t_hi) {
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  u32*& pos = FWD ? ws.f_pos : ws.r_pos;
  lefid_t*& ids = FWD ? ws.f_id : ws.r_id;
  u32* out_pos = ws.tmp[0];
  lefid_t* out_id = as_ids(ws.tmp[1]);
  u32* where_new = ws.tmp[7];
  const bool by_epoch_only = where == nullptr;
  if (ties) {
    // 4. order equal positions (epoch rule, then previous rank) with a stable odd-even
    //    transposition
    const u32 s_lo = t_lo > 0 ? t_lo - 1 : 0;
    const u32 s_hi = umin(n, t_hi + 2);  // slots [s_lo, s_hi)
    bool bad = true;
    while (bad) {
      bad = false;
      for (u32 parity = 0; parity < 2; ++parity) {
        for (u32 base = s_lo & ~1u; base < s_hi; base += 128) {
          const u32 k = base + 2 * lane + parity;
          bool sw = false;
          if (k >= s_lo && k + 1 < s_hi) {
            const u32 pa = out_pos[k], pb = out_pos[k + 1];
            if (pa == pb) {
              const u32 ia = out_id[k], ib = out_id[k + 1];
              bool ooo;
              if (by_epoch_only) {
                const u32 ea = ws.epoch[ia], eb = ws.epoch[ib];
                ooo = FWD ? ea < eb : ea > eb;
              } else {
                ooo = rank_pair_out_of_order<FWD>(ws, where, pa, ia, pb, ib);
              }
              if (ooo) {
                out_id[k] = ib;
                out_id[k + 1] = ia;
                if (!by_epoch_only) {
                  where_new[ib] = k;
                  where_new[ia] = k + 1;
                }
                sw = true;
              }
            }
          }
          bad = wave::any(sw) || bad;
        }
        wave::sync_mem();
      }
    }
  }
  // 5. the new arrays become current
  swap_ptr(pos, ws.tmp[0]);
  swap_with_scratch(ids, ws.tmp[1]);
  if (!by_epoch_only) {
    if (FWD) swap_ptr(ws.f_rank, ws.tmp[7]); else swap_ptr(ws.r_rank, ws.tmp[7]);
  }
  c.inv_valid[FWD ? 1 : 0] = !by_epoch_only;
}

// The rank update of the epoch loop when phase_bind_listed has left the keys of the units it bound
// (c.keys_valid): no split pass, and four consecutive ranks per lane.  The keys are sorted in LDS,
// then ONE sweep over the incoming rank order sends every carried-over unit to (its index among the
// carried-over units) + (keys before it) and notes, per key, how many carried-over units precede
// it; the new units follow from that.  Per block of 256 ranks: three 128-bit loads per lane, three
// cross-lane scans (running maximum of the carried-over positions, new units so far, keys so far)
// and four independent key searches per lane.
// Carried-over units that are out of order (a unit that went past another one behind an avoided
// secondary collision; every epoch has a few) are re-inserted like new units: the extrusion sweep
// of the previous epoch, which has the new positions in registers anyway, has marked them and
// listed their keys (a separate sweep over positions and marks used to find them here).
// Returns false -- nothing committed, the caller runs the general update -- when the keys do not
// fit the LDS buffers (RANK_KEY_CAP keys, the sort buffer less the sentinel; RANK_KEY_CAP_BIG in the
// generator's ring when that is free; 256 when the chromosome has 65536 LEFs or more and the per-key
// counts need 32 bits).
template <bool FWD>
MODLE_DEV_NOINLINE bool rank_update_listed(Cell& c) {
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 n_listed = wave::uniform(c.n_keys);
  const u32 lane = wave::lane();
  const u32* pos = FWD ? ws.f_pos : ws.r_pos;
  const lefid_t* ids = FWD ? ws.f_id : ws.r_id;
  const move_t* marks = FWD ? ws.f_move : ws.r_move;
#ifdef MODLE_SUBTIMER_RANK
  const u64 t_enter = wave::clock();
#endif
  // per key: the number of carried-over units that go before it.  With fewer than 65536 LEFs -- every
  // real chromosome -- the count rides in the key itself, (position, previous rank, count) =
  // 32 + 16 + 16 bits: one LDS word per key, nothing beside it, and the comparisons of the sweep are
  // unchanged because no two keys share (position, previous rank).  Otherwise: 32-bit counts in the
  // staging buffer, and STAGE_CAP keys.
  const bool narrow = n < 65536u;
  const u32 kshift = narrow ? 16u : 0u;
  u32* cnt32 = c.lds.stage;
  // the keys: the sort buffer (RANK_KEY_CAP).  An epoch that re-inserts more (the collision-heavy
  // configurations: 600-800 units per epoch and direction on the large chromosomes) borrows the 8 KB
  // of the generator's ring, whose contents wait in device memory meanwhile -- when the ring is this
  // wave's to borrow (helper-wave mode lends it to the helper for the burn-in epochs)
  const u32 nd_listed = wave::uniform(c.n_disp[FWD ? 1 : 0]);
  const bool big = n_listed + nd_listed > RANK_KEY_CAP;
  if (big && (!narrow || c.ring_lent || n_listed + nd_listed > RANK_KEY_CAP_BIG ||
              ws.capacity_lefs < RING_SPILL_AT + RNG_RING))
    return false;
  const u32 key_cap = big ? RANK_KEY_CAP_BIG : (narrow ? RANK_KEY_CAP : STAGE_CAP);
  u64* const keys = big ? c.g.ring : c.lds.sort_lds;
  u64* const ring_spill = ws.sort_keys + RING_SPILL_AT;
  if (big) {
    for (u32 k = lane; k < RNG_RING; k += 64) ring_spill[k] = c.g.ring[k];
  }
  const auto give_ring_back = [&]() {
    if (big) {
      wave::sync_mem();
      for (u32 k = lane; k < RNG_RING; k += 64) c.g.ring[k] = ring_spill[k];
      wave::sync_lds();
    }
  };
  const auto in_lds_format = [&](u64 kv) -> u64 {
    return (kv & 0xFFFFFFFF00000000ull) | (static_cast<u64>(static_cast<u32>(kv)) << kshift);
  };
  const auto cnt_load = [&](u32 q, u64 key) -> u32 { return narrow ? static_cast<u32>(key) & 0xFFFFu : cnt32[q]; };
  const u64* src = reinterpret_cast<const u64*>(FWD ? ws.tmp[4] : ws.tmp[2]);
  u32* out_pos = ws.tmp[0];
  lefid_t* out_id = as_ids(ws.tmp[1]);
  const u32 nblk = (n + 255) / 256;
  wave::lockstep();
  for (u32 base = 0; base < n_listed; base += 64) {
    const u32 k = base + lane;
    const u64 kv = wave::ld_sel(src, k, k < n_listed, ~u64(0));
    if (k < n_listed) keys[k] = in_lds_format(kv);
  }
  // the out-of-order units the extrusion sweep listed, unless they have been released and bound
  // again since (their slot then carries the mark of a new unit, and the bind phase's key)
  u32 n_new = n_listed;
  {
    const u64* dsrc = reinterpret_cast<const u64*>(FWD ? ws.tmp[7] : ws.tmp[6]);
    const u32 nd = nd_listed;
    for (u32 base = 0; base < nd; base += 64) {
      const u32 e = base + lane;
      const u64 kv = wave::ld_sel(dsrc, e, e < nd, ~u64(0));
      const bool still = e < nd && wave::ld_sel(marks, static_cast<u32>(kv), e < nd, 0u) == DISP_MARK;
      const u64 dm = wave::ballot(still);
      const u32 j = n_new + static_cast<u32>(wave::popc64(dm & lanemask_lt(lane)));
      if (still && j < key_cap) keys[j] = in_lds_format(kv);
      n_new += static_cast<u32>(wave::popc64(dm));
    }
  }
#ifdef MODLE_EMU_TRACE_RANK
  if (n_new > key_cap && lane == 0) fprintf(stderr, "rank_update_listed: n_new %u > key_cap %u: general update\n", n_new, key_cap);
#endif
  if (n_new > key_cap) {
    give_ring_back();
    return false;
  }
#ifdef MODLE_EMU_TRACE_RANK  // (emulator only: which regime a test exercises)
  if (lane == 0) fprintf(stderr, "rank_update_listed: %s n_new %u (listed %u, displaced %u) of %u, key_cap %u\n", FWD ? "fwd" : "rev", n_new, n_listed, n_new - n_listed, n, key_cap);
#endif
  const u32 n_old = n - n_new;
  const u32 m2 = n_new != 0 ? pow2_ceil(n_new) : 0;
  for (u32 k = n_new + lane; k < m2; k += 64) keys[k] = ~u64(0);
  // (a key of all ones behind the last one, also when n_new is a power of two: a step of the
  // searches below that overshoots reads it, through one `v_min` on the index, instead of testing
  // its range; n_new <= key_cap, one less than the buffer holds)
  if (lane == 0) keys[n_new] = ~u64(0);
  // (the count every key starts from: no carried-over unit follows it.  Key j was written by whichever
  // lane listed it: the read-modify-write below is another lane's, so the writes have to be complete)
  wave::sync_lds();
  if (narrow) {
    for (u32 j = lane; j < n_new; j += 64) keys[j] |= n_old;
  } else {
    for (u32 j = lane; j < n_new; j += 64) cnt32[j] = n_old;
  }
  wave::sync_lds();
  if (m2 > 1) bitonic_sort_u64<true>(keys, m2);
#ifdef MODLE_SUBTIMER_RANK
  c.ph[14] += wave::clock() - t_enter;  // (keys: load, displaced units, sort)
  const u64 t_sweep = wave::clock();
#endif

  bool ties = false;
  u32 t_lo = 0xFFFFFFFFu, t_hi = 0;  // output slots of the units flagged for equal positions
  u32 seen_new = 0;   // new units in the blocks before this one
  u32 run_max = 0;    // max position of the carried-over units before this block
  u32 carry_lo = 0;   // keys before the last carried-over unit so far
  struct Blk {
    wave::U32x4 P, I, K;
  };
  const auto load_blk = [&](u32 t, Blk& r) {
    const u32 w = 256 * t + 4 * lane;
    const u32 wq = w < n ? w : 0u;
    r.P = wave::ld4(pos, wq);
    r.I = wave::ld4(ids, wq);
    r.K = wave::ld4(marks, wq);
  };
  // (the block's registers are taken over at the BOTTOM of the loop, behind the stores: there the
  // compiler can count what was issued after the loads and waits for the loads alone; at the top,
  // where the first iteration and the back edge meet, it would wait for the stores as well)
  Blk cur;
  load_blk(0, cur);
  Blk g = cur;
  for (u32 t = 0; t < nblk; ++t) {
    if (t + 1 < nblk) load_blk(t + 1, cur);
    const u32 w = 256 * t + 4 * lane;
    u32 pp[4], oid[4], mx[4], nb[4];
    bool carried[4];  // here: carried over AND still in order (the units that keep their order)
    bool act4[4];
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      act4[j] = w + j < n;
      pp[j] = g.P.v[j];
      oid[j] = g.I.v[j];
      carried[j] = act4[j] && g.K.v[j] != NEW_MARK && g.K.v[j] != DISP_MARK;
      const u32 cp = carried[j] ? pp[j] : 0u;
      mx[j] = j == 0 ? cp : umax(mx[j - 1], cp);  // running max of the carried-over positions
    }
    const u32 pm = wave_prefix_max_u32(mx[3]);
    const u32 pm_prev = wave::shfl_up1(pm);
    const u32 lane_excl = umax(run_max, lane > 0 ? pm_prev : 0);
    run_max = umax(run_max, wave::bcast(pm, 63));
    u32 excl[4];  // position of the carried-over unit before unit j (0: none)
    u32 lane_new = 0;
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      excl[j] = j == 0 ? lane_excl : umax(lane_excl, mx[j - 1]);
      // (every out-of-order unit carries DISP_MARK: the extrusion sweep compares against ALL units
      // of lower rank, this maximum runs over fewer.  Should one slip through, the count at the end
      // does not add up and the general update takes over.)
      carried[j] = carried[j] && !(pp[j] < excl[j]);
      nb[j] = lane_new;  // re-inserted units of this lane before unit j
      lane_new += (act4[j] && !carried[j]) ? 1u : 0u;
    }
    const u32 ps = wave_prefix_sum_u32(lane_new);
    const u32 lane_before = seen_new + ps - lane_new;
    seen_new += wave::bcast(ps, 63);
    // lo = number of keys that go before the unit (see rank_merge): four searches side by side
    u32 lo[4];
    u64 thr[4];
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      lo[j] = carried[j] ? carry_lo : 0u;
      // (position, previous rank): units and keys with equal positions merge in the order of their
      // previous ranks, which is what lets rank_finish order them by binding epoch alone; a unit
      // that takes no part has the key nothing lies below
      thr[j] = carried[j] ? (static_cast<u64>(pp[j]) << 32) | ((w + j) << kshift) : u64(0);
    }
#pragma unroll
    for (u32 sft = 8; sft >= 1; sft >>= 1) {
      // (the four reads of a round are issued together: left alone the compiler waits for each)
      u32 jx[4];
      u64 kv[4];
#pragma unroll
      for (u32 j = 0; j < 4; ++j) {
        jx[j] = lo[j] + sft;
        kv[j] = (keys - 1)[umin(jx[j], n_new + 1)];  // (beyond the keys: the sentinel)
      }
      wave::sched_fence();
#pragma unroll
      for (u32 j = 0; j < 4; ++j) {
        if (kv[j] < thr[j]) lo[j] = jx[j];
      }
      wave::sched_fence();
    }
    bool far = false;  // the fixed steps ran out: finish with a binary search (rare)
#pragma unroll
    for (u32 j = 0; j < 4; ++j) far = far || (carried[j] && lo[j] == carry_lo + 15 && lo[j] < n_new);
    if (wave::any(far)) {
#pragma unroll
      for (u32 j = 0; j < 4; ++j) {
        if (carried[j] && lo[j] == carry_lo + 15 && lo[j] < n_new) {
          u32 hi = n_new;
          u32 l = lo[j];
          while (l < hi) {
            const u32 mid = (l + hi) >> 1;
            if (keys[mid] < thr[j]) l = mid + 1; else hi = mid;
          }
          lo[j] = l;
        }
      }
    }
    u32 lmx[4];
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      const u32 cl = carried[j] ? lo[j] : 0u;
      lmx[j] = j == 0 ? cl : umax(lmx[j - 1], cl);  // keys before the carried-over units so far
    }
    const u32 lpm = wave_prefix_max_u32(lmx[3]);
    const u32 lpm_prev = wave::shfl_up1(lpm);
    const u32 lane_lo = umax(carry_lo, lane > 0 ? lpm_prev : 0);
    carry_lo = umax(carry_lo, wave::bcast(lpm, 63));
    bool tie = false;
    u32 tie_lo = 0xFFFFFFFFu, tie_hi = 0;
    u32 slot[4], lo_prev[4];
    bool gaps = false;  // keys lie between a unit and the carried-over unit before it
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      slot[j] = w + j - (lane_before + nb[j]) + lo[j];
      lo_prev[j] = j == 0 ? lane_lo : umax(lane_lo, lmx[j - 1]);
      gaps = gaps || (carried[j] && lo_prev[j] < lo[j]);
    }
    if (wave::any(gaps)) {
#pragma unroll
      for (u32 j = 0; j < 4; ++j) {
        if (carried[j]) {
          // keys [lo of the carried-over unit before, lo) lie between that unit and this one; a
          // key at the position of either neighbour is flagged for the final ordering
          const u32 a = slot[j] - lo[j];
          for (u32 q = lo_prev[j]; q < lo[j]; ++q) {
            const u64 kq = keys[q];
            if (narrow) keys[q] = (kq & ~u64(0xFFFF)) | a; else cnt32[q] = a;
            const u32 kp = static_cast<u32>(kq >> 32);
            if (kp != UNBOUND && (kp == pp[j] || (a > 0 && kp == excl[j]))) {
              tie = true;
              tie_lo = umin(tie_lo, q + a);
              tie_hi = umax(tie_hi, q + a);
            }
          }
        }
      }
    }
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      const u32 a = slot[j] - lo[j];
      const bool tj = carried[j] && pp[j] != UNBOUND && a > 0 && excl[j] == pp[j];
      if (tj) {
        tie = true;
        tie_lo = umin(tie_lo, slot[j]);
        tie_hi = umax(tie_hi, slot[j]);
      }
      // (unconditional stores: the lanes that have nothing to store hit a scratch word.  With the
      // stores under a branch the compiler cannot count them, and the wait for the next block's
      // loads at the top of the loop becomes a wait for these stores as well)
      u32* const dump = reinterpret_cast<u32*>(ws.sort_keys) + lane;
      *(carried[j] ? &out_pos[slot[j]] : dump) = pp[j];
      *(carried[j] ? &out_id[slot[j]] : reinterpret_cast<lefid_t*>(dump)) = static_cast<lefid_t>(oid[j]);
    }
    if (wave::any(tie)) {
      ties = true;
      t_lo = umin(t_lo, ~wave::bcast(wave_prefix_max_u32(~tie_lo), 63));
      t_hi = umax(t_hi, wave::bcast(wave_prefix_max_u32(tie_hi), 63));
    }
    // (two blocks of loads in flight were tried in round 4, the proper way -- three register sets, the
    // loop unrolled by two, no set copied while its loads are on their way: 3 % SLOWER on the default
    // launch (19 spilled VGPRs instead of 11, twice the code): the sweep is bound by its ~500
    // instructions and four dependent search rounds per block, not by the latency of its loads)
    if (t + 1 < nblk) g = cur;
  }
  if (seen_new != n_new) {  // (the marks and the list disagree: cannot happen)
    give_ring_back();
    return false;
  }
  wave::sync_lds();
#ifdef MODLE_SUBTIMER_RANK
  c.ph[15] += wave::clock() - t_sweep;  // (the sweep)
#endif
  for (u32 base = 0; base < n_new; base += 64) {
    const u32 bq = base + lane;
    bool tie = false;
    if (bq < n_new) {
      const u64 key = keys[bq];
      const u32 pp = static_cast<u32>(key >> 32);
      const u32 lo = cnt_load(bq, key);
      const u32 nid = ids[static_cast<u32>(key) >> kshift];  // the slot the unit was bound in
      wave::st_stream(&out_pos[bq + lo], pp);
      wave::st_stream(&out_id[bq + lo], nid);
      tie = bq + 1 < n_new && static_cast<u32>(keys[bq + 1] >> 32) == pp;
      // (a key behind the last carried-over unit, at its position: no unit follows to flag it)
      tie = tie || (lo == n_old && n_old != 0 && pp == run_max);
      tie = tie && pp != UNBOUND;
    }
    if (wave::any(tie)) {
      ties = true;
      const u32 slot = bq < n_new ? bq + cnt_load(bq, keys[bq]) : 0;
      t_lo = umin(t_lo, ~wave::bcast(wave_prefix_max_u32(tie ? ~slot : 0u), 63));
      t_hi = umax(t_hi, wave::bcast(wave_prefix_max_u32(tie ? slot : 0u), 63));
    }
  }
  give_ring_back();
  wave::sync_mem();
  rank_finish<FWD>(c, ties, nullptr, t_lo, t_hi);
  return true;
}

// all_new: treat every entry as newly bound (full sort; used by the phase-level test entry point)
template <bool FWD>
MODLE_DEV_NOINLINE void rank_update(Cell& c, bool all_new) {
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  if (n < 2) return;
  {
    const bool listed = !all_new && c.keys_valid && c.disp_valid && c.n_keys <= RANK_KEY_CAP_BIG;
#ifdef MODLE_EMU_TRACE_RANK
    if (!listed && wave::lane() == 0) fprintf(stderr, "rank_update: general (all_new %d keys_valid %d disp_valid %d n_keys %u n_disp %u %u)\n", int(all_new), int(c.keys_valid), int(c.disp_valid), c.n_keys, c.n_disp[0], c.n_disp[1]);
#endif
    if (listed && rank_update_listed<FWD>(c)) {
      if (FWD) c.keys_valid = false;  // (the keys serve the rev update, then the fwd update)
      return;
    }
    c.keys_valid = false;  // (the general update below overwrites the arrays that hold them)
  }
  ensure_inverse<FWD>(c);  // the previous ranks by LEF id are the last tie-break
  const u32 lane = wave::lane();
  const u32* pos = FWD ? ws.f_pos : ws.r_pos;
  const lefid_t* ids = FWD ? ws.f_id : ws.r_id;
  const move_t* marks = FWD ? ws.f_move : ws.r_move;
  u32* where = FWD ? ws.f_rank : ws.r_rank;  // previous ranks until the final scatter
  u32* old_pos = ws.tmp[2];
  lefid_t* old_id = as_ids(ws.tmp[3]);
  lefid_t* new_id = as_ids(ws.tmp[4]);
  u64* keys_lds = c.lds.sort_lds;
  u64* keys_glb = ws.sort_keys;

  // 1. stable split.  A carried-over unit that is no longer in order (its position is below the
  //    running maximum of the carried-over units before it; this can happen after
  //    fix_secondary_lef_lef_collisions re-positions a pair) is handled like a new unit, so that
  //    the kept sequence is non-decreasing by construction.
  u32 n_old = 0, n_new = 0;
  u32 run_max = 0;  // max position of carried-over units in previous batches

  constexpr u32 UX = 4;  // batches per group; the next group's loads go before this group's stores
  struct UnitRegs {
    u32 P[UX], I[UX], K[UX];
  };
  const auto load_units = [&](auto op, u32 group, UnitRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 kq = group + 64 * u + lane;
      r.P[u] = op(pos, kq, kq < n, 0, r.P[u]);
      r.I[u] = op(ids, kq, kq < n, 0, r.I[u]);
      r.K[u] = op(marks, kq, kq < n, 0, r.K[u]);
    }
  };
  UnitRegs cur;
  load_units(wave::LdRaw{}, 0, cur);
  for (u32 group = 0; group < n; group += 64 * UX) {
    UnitRegs g = cur;
    load_units(wave::LdMask{}, group, g);  // (defaults of the lanes outside the range)
    if (group + 64 * UX < n) load_units(wave::LdRaw{}, group + 64 * UX, cur);
    const u32* Pq = g.P;
    const u32* Iq = g.I;
    const u32* Kq = g.K;
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
    const u32 base = group + 64 * u;
    if (base >= n) break;
    const u32 k = base + lane;
    const bool act = k < n;
    const u32 P = Pq[u];
    const u32 id = Iq[u];
    const bool fresh = act && (all_new || Kq[u] == NEW_MARK);
    const bool carried = act && !fresh;
    const u32 pm = wave_prefix_max_u32(carried ? P : 0);
    const u32 incl_last = wave::bcast(pm, 63);
    const u32 pm_prev = wave::shfl_up1(pm);
    const u32 excl = umax(run_max, lane > 0 ? pm_prev : 0);
    const bool displaced = carried && P < excl;
    const bool is_new = fresh || displaced;
    const bool is_old = carried && !displaced;
    const u64 mn = wave::ballot(is_new);
    const u64 mo = wave::ballot(is_old);
    if (is_new) {
      const u32 j = n_new + static_cast<u32>(wave::popc64(mn & lanemask_lt(lane)));
      new_id[j] = id;
      const u64 key = (static_cast<u64>(P) << 32) | j;
      if (j < SORT_LDS_CAP) keys_lds[j] = key; else keys_glb[j] = key;
    }
    if (is_old) {
      const u32 j = n_old + static_cast<u32>(wave::popc64(mo & lanemask_lt(lane)));
      wave::st_stream(&old_id[j], id);
      wave::st_stream(&old_pos[j], P);
    }
    n_new += static_cast<u32>(wave::popc64(mn));
    n_old += static_cast<u32>(wave::popc64(mo));
    run_max = umax(run_max, incl_last);
    }
  }
  wave::sync_mem();
  if (n_old + n_new != n) {
    c.error = ERR_INTERNAL;  // cannot happen: every active unit is either carried over or new
    return;
  }
  // 2. sort the new units by (position, previous rank): in LDS, or in device memory when there
  //    are more of them than the LDS buffer holds (whole-chromosome rebinding only)
  if (n_new != 0) {
    const u32 m2 = pow2_ceil(n_new);
    if (n_new <= SORT_LDS_CAP) {
      for (u32 base = n_new; base < m2; base += 64) {
        const u32 k = base + lane;
        if (k < m2) keys_lds[k] = ~u64(0);
      }
      wave::sync_lds();
      if (m2 > 1) bitonic_sort_u64<true>(keys_lds, m2);
    } else {
      for (u32 base = 0; base < SORT_LDS_CAP; base += 64) keys_glb[base + lane] = keys_lds[base + lane];
      for (u32 base = n_new; base < m2; base += 64) {
        const u32 k = base + lane;
        if (k < m2) keys_glb[k] = ~u64(0);
      }
      wave::sync_mem();
      bitonic_sort_u64<false>(keys_glb, m2);
    }
  }
  // 3. merge by cross-ranking (kept units are sorted) straight into the output arrays and the
  //    new inverse permutation.  Equal positions of bound units are the only thing this does not
  //    order completely (epoch rule); they are rare, so they are only flagged here.
  u32* out_pos = ws.tmp[0];
  lefid_t* out_id = as_ids(ws.tmp[1]);
  u32* where_new = ws.tmp[7];
  const bool ties = (n_new <= SORT_LDS_CAP)
                        ? rank_merge<FWD>(keys_lds, n_new, n_old, old_pos, old_id, new_id, out_pos,
                                          out_id, where_new, c.lds.stage)
                        : rank_merge<FWD>(keys_glb, n_new, n_old, old_pos, old_id, new_id, out_pos,
                                          out_id, where_new, c.lds.stage);
  wave::sync_mem();
  rank_finish<FWD>(c, ties, where, 0, n - 1);
}

}  // namespace modle_dev

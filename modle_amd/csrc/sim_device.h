// sim_device.h -- one wavefront simulates one cell: the per-epoch loop of
// Simulation::simulate_one_cell (reference: src/libmodle/cpu/simulation.cpp:896-986) written for
// a 64-lane wave.  Included after a `wave` backend (wave_hip.h on the GPU).
//
// Data layout (sim_types.h: Workspace).  Extrusion units are kept in RANK ORDER, rev and fwd
// units separately: r_pos[k] / r_move[k] / r_coll[k] / r_id[k] describe the k-th rev unit in
// 5'->3' order.  Every pass that walks units in genomic order -- move adjustment, all collision
// passes, extrusion -- therefore streams contiguous memory.  The things the reference does in
// LEF-id order because of the PRNG draw order (move generation, release, bind) use id-ordered
// arrays and cross over with one scatter through the inverse permutations r_rank / f_rank.
// LEF-LEF collision words carry LEF ids like the reference's; barrier collision words carry the
// barrier index.
#pragma once
#include "sim_rng.h"

namespace modle_dev {

struct Cell {
  const Params* p;
  const Interval* iv;
  Workspace ws;
  WaveLds lds;
  Rng g;
  u32 n_lefs;     // Task::num_lefs
  u32 n_active;   // State::num_active_lefs
  u32 hist_len;   // entries in the burn-in history buffers
  u32 hist_head;  // ring head
  u32 error;      // non-zero when an internal capacity was exceeded (uniform)
  u32 n_hit[2];   // entries of ws.hit_pos / hit_idx (stalling barriers of this epoch; uniform)
  // LEFs released by release_lefs, in LEF-id order, listed in LDS (lds.sort_lds as REL_CAP
  // words) for the next epoch's select_and_bind_lefs; rel_valid = the list is complete
  u32 n_rel;
  bool rel_valid;
  u32 n_bound;    // LEFs [0, n_bound) have been bound at least once (the rest were just activated)
  // phase_bind_listed leaves the sort keys of the units it bound ((position << 32) | rank slot, one
  // set per direction, in ws.tmp[2..3] / ws.tmp[4..5]) for the two rank updates that follow it
  u32 n_keys;
  bool keys_valid;
  // the extrusion sweep lists the units it leaves out of order (their position after the move is
  // below that of a unit of lower rank: a unit went past another one behind an avoided secondary
  // collision): sort keys in ws.tmp[6] (rev) / ws.tmp[7] (fwd), DISP_MARK in the move array.  The
  // rank update re-inserts them like the units bound in between.
  u32 n_disp[2];
  bool disp_valid;
  // ws.r_rank / ws.f_rank ([0] rev, [1] fwd) hold the complete inverse permutation.  The rank
  // update of the epoch loop does not write it (one scattered store per unit and epoch): the
  // sparse consumers -- bind, release, fix_secondary -- get the ranks of the few LEFs they need
  // from sweeps that pass over the id arrays anyway (RankFilter below), everything else
  // (contact sampling, the general rank update, the phase-level hooks) calls ensure_inverse.
  bool inv_valid[2];
  // the secondary pass collects the LEFs of its avoided collisions in the LDS id filter (for the
  // rank lookups of fix_secondary)
  bool filter_on;
#ifdef MODLE_PHASE_TIMERS
  u64 ph[16];     // profiling build: time spent per phase (wave::clock ticks)
#endif
};
// Profiling build (make prof): PHASE(c, i, call) accumulates the time of `call` in c.ph[i].
#ifdef MODLE_PHASE_TIMERS
#define PHASE(c, i, ...)                          \
  do {                                            \
    const u64 ph_t0_ = wave::clock();             \
    __VA_ARGS__;                                  \
    (c).ph[i] += wave::clock() - ph_t0_;          \
  } while (0)
#else
#define PHASE(c, i, ...) \
  do {                   \
    __VA_ARGS__;         \
  } while (0)
#endif
constexpr u32 REL_CAP = 2 * SORT_LDS_CAP;  // u32 entries in the LDS sort buffer
constexpr u32 ERR_LIST_OVERFLOW = 1;
constexpr u32 ERR_TRIAL_OVERFLOW = 2;
constexpr u32 ERR_INTERNAL = 3;
constexpr u32 ERR_CANCELLED = 4;  // the host raised the abort word (reference: _ctx polled per epoch)

template <class T>
MODLE_DEV void swap_ptr(T*& a, T*& b) {
  T* t = a;
  a = b;
  b = t;
}

// first barrier index whose position is >= key
MODLE_DEV u32 bar_lower_bound(const Interval& iv, u64 key) {
  const u32 nb = iv.n_barriers;
  if (key <= iv.start) return 0;
  const u64 b = (key - iv.start) >> iv.bucket_shift;
  if (b >= iv.n_buckets) return nb;
  u32 i = iv.bar_bucket[b];
  while (i < nb && iv.bar_pos[i] < key) ++i;
  return i;
}

MODLE_DEV u32 lower_bound_u32(const u32* a, u32 n, u32 key) {  // first index with a[i] >= key
  u32 lo = 0, hi = n;
  while (lo < hi) {
    const u32 mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Rebuilds the inverse permutation of one direction from the id array (one scattered store per
// unit: only where the complete permutation is really needed).
template <bool FWD>
MODLE_DEV_NOINLINE void ensure_inverse(Cell& c) {
  if (c.inv_valid[FWD ? 1 : 0]) return;
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  const u32* ids = FWD ? ws.f_id : ws.r_id;
  u32* rank = FWD ? ws.f_rank : ws.r_rank;
  const u32 nblk = (n + 255) / 256;
  for (u32 t = 0; t < nblk; ++t) {
    const u32 w = 256 * t + 4 * lane;
    const wave::U32x4 I = wave::ld4(ids, w < n ? w : 0u);
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      if (w + q < n) rank[I.v[q]] = w + q;
    }
  }
  wave::sync_mem();
  c.inv_valid[FWD ? 1 : 0] = true;
}
MODLE_DEV void ensure_inverse_both(Cell& c) {
  ensure_inverse<false>(c);
  ensure_inverse<true>(c);
}

// A set of LEF ids as a bitmap in LDS (the sort buffer, idle outside the rank update and the
// collision passes that stage windows there): RANK_FILTER_BITS bits indexed by id modulo that
// size.  Up to 32768 LEFs the test is exact; beyond, ids that share a bit with a member pass as
// well, which only costs the sweeps that use the filter a few useless stores.
constexpr u32 RANK_HARD = 0x80000000u;  // flag on a rank reported by the extrusion sweep: hard stall
constexpr u32 RANK_FILTER_WORDS = SORT_LDS_CAP;  // 64-bit words
constexpr u32 RANK_FILTER_BITS = 64 * RANK_FILTER_WORDS;
MODLE_DEV void rank_filter_clear(Cell& c, u32 n_ids) {
  u64* bm = c.lds.sort_lds;
  const u32 nw = umin(RANK_FILTER_WORDS, (n_ids + 63) / 64);
  wave::lockstep();
  for (u32 k = wave::lane(); k < nw; k += 64) bm[k] = 0;
  wave::sync_lds();
}
// adds the ids [first, first + 64) whose bit is set in `members` (uniform)
MODLE_DEV void rank_filter_add_mask(Cell& c, u32 first, u64 members) {
  u64* bm = c.lds.sort_lds;
  if (wave::lane() == 0) bm[(first / 64) % RANK_FILTER_WORDS] |= members;
}
// adds the id of the calling lane (any subset of the lanes may call)
MODLE_DEV void rank_filter_add_id(Cell& c, u32 id) {
  u32* bm = reinterpret_cast<u32*>(c.lds.sort_lds);
  wave::lds_or_u32(&bm[(id % RANK_FILTER_BITS) >> 5], 1u << (id & 31u));
}
MODLE_DEV bool rank_filter_test(const Cell& c, u32 id) {
  const u32* bm = reinterpret_cast<const u32*>(c.lds.sort_lds);
  return ((bm[(id % RANK_FILTER_BITS) >> 5] >> (id & 31u)) & 1u) != 0;
}

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
                          const u32* old_id, const u32* new_id, u32* out_pos, u32* out_id,
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
  u32*& ids = FWD ? ws.f_id : ws.r_id;
  u32* out_pos = ws.tmp[0];
  u32* out_id = ws.tmp[1];
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
  swap_ptr(ids, ws.tmp[1]);
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
// fit the LDS buffers.
template <bool FWD>
MODLE_DEV_NOINLINE bool rank_update_listed(Cell& c) {
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 n_listed = wave::uniform(c.n_keys);
  const u32 lane = wave::lane();
  const u32* pos = FWD ? ws.f_pos : ws.r_pos;
  const u32* ids = FWD ? ws.f_id : ws.r_id;
  const u32* marks = FWD ? ws.f_move : ws.r_move;
  u64* keys = c.lds.sort_lds;
  u32* cnt_lds = c.lds.stage;
  const u64* src = reinterpret_cast<const u64*>(FWD ? ws.tmp[4] : ws.tmp[2]);
  u32* out_pos = ws.tmp[0];
  u32* out_id = ws.tmp[1];
  const u32 nblk = (n + 255) / 256;
  wave::lockstep();
  for (u32 base = 0; base < n_listed; base += 64) {
    const u32 k = base + lane;
    const u64 kv = wave::ld_sel(src, k, k < n_listed, ~u64(0));
    if (k < n_listed) keys[k] = kv;
  }
  // the out-of-order units the extrusion sweep listed, unless they have been released and bound
  // again since (their slot then carries the mark of a new unit, and the bind phase's key)
  u32 n_new = n_listed;
  {
    const u64* dsrc = reinterpret_cast<const u64*>(FWD ? ws.tmp[7] : ws.tmp[6]);
    const u32 nd = wave::uniform(c.n_disp[FWD ? 1 : 0]);
    for (u32 base = 0; base < nd; base += 64) {
      const u32 e = base + lane;
      const u64 kv = wave::ld_sel(dsrc, e, e < nd, ~u64(0));
      const bool still = e < nd && wave::ld_sel(marks, static_cast<u32>(kv), e < nd, 0u) == DISP_MARK;
      const u64 dm = wave::ballot(still);
      const u32 j = n_new + static_cast<u32>(wave::popc64(dm & lanemask_lt(lane)));
      if (still && j < STAGE_CAP) keys[j] = kv;
      n_new += static_cast<u32>(wave::popc64(dm));
    }
  }
  if (n_new > STAGE_CAP) return false;
  const u32 n_old = n - n_new;
  const u32 m2 = n_new != 0 ? pow2_ceil(n_new) : 0;
  for (u32 k = n_new + lane; k < m2; k += 64) keys[k] = ~u64(0);
  for (u32 j = lane; j < n_new; j += 64) cnt_lds[j] = n_old;
  wave::sync_lds();
  if (m2 > 1) bitonic_sort_u64<true>(keys, m2);

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
      // previous ranks, which is what lets rank_finish order them by binding epoch alone
      thr[j] = (static_cast<u64>(pp[j]) << 32) | (w + j);
    }
#pragma unroll
    for (u32 sft = 8; sft >= 1; sft >>= 1) {
      // (the four reads of a round are issued together: left alone the compiler waits for each)
      u32 jx[4];
      bool in[4];
      u64 kv[4];
#pragma unroll
      for (u32 j = 0; j < 4; ++j) {
        jx[j] = lo[j] + sft;
        in[j] = carried[j] & (jx[j] <= n_new);
        kv[j] = keys[in[j] ? jx[j] - 1 : 0];  // (no branch around the read)
      }
      wave::sched_fence();
#pragma unroll
      for (u32 j = 0; j < 4; ++j) {
        if (in[j] & (kv[j] < thr[j])) lo[j] = jx[j];
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
            cnt_lds[q] = a;
            const u32 kp = static_cast<u32>(keys[q] >> 32);
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
      *(carried[j] ? &out_id[slot[j]] : dump) = oid[j];
    }
    if (wave::any(tie)) {
      ties = true;
      t_lo = umin(t_lo, ~wave::bcast(wave_prefix_max_u32(~tie_lo), 63));
      t_hi = umax(t_hi, wave::bcast(wave_prefix_max_u32(tie_hi), 63));
    }
    if (t + 1 < nblk) g = cur;
  }
  if (seen_new != n_new) return false;  // (the marks and the list disagree: cannot happen)
  wave::sync_lds();
  for (u32 base = 0; base < n_new; base += 64) {
    const u32 bq = base + lane;
    bool tie = false;
    if (bq < n_new) {
      const u64 key = keys[bq];
      const u32 pp = static_cast<u32>(key >> 32);
      const u32 lo = cnt_lds[bq];
      const u32 nid = ids[static_cast<u32>(key)];  // the slot the unit was bound in
      wave::st_stream(&out_pos[bq + lo], pp);
      wave::st_stream(&out_id[bq + lo], nid);
      tie = bq + 1 < n_new && static_cast<u32>(keys[bq + 1] >> 32) == pp;
      // (a key behind the last carried-over unit, at its position: no unit follows to flag it)
      tie = tie || (lo == n_old && n_old != 0 && pp == run_max);
      tie = tie && pp != UNBOUND;
    }
    if (wave::any(tie)) {
      ties = true;
      const u32 slot = bq < n_new ? bq + cnt_lds[bq < n_new ? bq : 0] : 0;
      t_lo = umin(t_lo, ~wave::bcast(wave_prefix_max_u32(tie ? ~slot : 0u), 63));
      t_hi = umax(t_hi, wave::bcast(wave_prefix_max_u32(tie ? slot : 0u), 63));
    }
  }
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
    const bool listed = !all_new && c.keys_valid && c.disp_valid && c.n_keys <= STAGE_CAP;
    if (listed && rank_update_listed<FWD>(c)) {
      if (FWD) c.keys_valid = false;  // (the keys serve the rev update, then the fwd update)
      return;
    }
    c.keys_valid = false;  // (the general update below overwrites the arrays that hold them)
  }
  ensure_inverse<FWD>(c);  // the previous ranks by LEF id are the last tie-break
  const u32 lane = wave::lane();
  const u32* pos = FWD ? ws.f_pos : ws.r_pos;
  const u32* ids = FWD ? ws.f_id : ws.r_id;
  const u32* marks = FWD ? ws.f_move : ws.r_move;
  u32* where = FWD ? ws.f_rank : ws.r_rank;  // previous ranks until the final scatter
  u32* old_pos = ws.tmp[2];
  u32* old_id = ws.tmp[3];
  u32* new_id = ws.tmp[4];
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
  u32* out_id = ws.tmp[1];
  u32* where_new = ws.tmp[7];
  const bool ties = (n_new <= SORT_LDS_CAP)
                        ? rank_merge<FWD>(keys_lds, n_new, n_old, old_pos, old_id, new_id, out_pos,
                                          out_id, where_new, c.lds.stage)
                        : rank_merge<FWD>(keys_glb, n_new, n_old, old_pos, old_id, new_id, out_pos,
                                          out_id, where_new, c.lds.stage);
  wave::sync_mem();
  rank_finish<FWD>(c, ties, where, 0, n - 1);
}

// =============================================================================================
// generate_moves (reference: simulation.cpp:272-330).  Draws are made in LEF-id order (the
// reference's stream order) and scattered to the unit's slot in rank order.
// =============================================================================================
MODLE_DEV u32 move_from_normal(f64 unit, f64 speed, f64 std) {
  const f64 v = unit * std + speed;
  return static_cast<u32>(static_cast<u64>(wave::f_round(v > 0.0 ? v : 0.0)));
}

// Queue of drawn moves (LDS, c.lds.stage): entry e lives at slot e % MOVQ_CAP, its move in the
// first half of the buffer and the low word of the stream position right after its draw in the
// second half.
constexpr u32 MOVQ_CAP = STAGE_CAP / 2;

// One step of the draw stream of generate_moves: lane l evaluates the normal-distribution attempt
// that would start at stream position pos + l (Boost's ziggurat, sim_rng.h: unit_normal_exact).
// An attempt takes one raw output (the strip's rectangle, ~98.8 %) or two (wedge test: accepted
// or rejected); which positions really start an attempt follows from the chain "an attempt that
// takes two outputs hides the position after it".  Accepted attempts are appended to the queue in
// stream order; the rare attempts whose length is data dependent beyond that (tail of the
// distribution, a uniform_01 retry) are replayed by the sequential routine.  Returns the new
// queue tail; uniform.
MODLE_DEV u32 draw_moves_step(Cell& c, f64 speed, f64 std, u32 tail) {
  const u32 lane = wave::lane();
  Rng& g = c.g;
  u32* q_move = c.lds.stage;
  u32* q_end = c.lds.stage + MOVQ_CAP;
  rng_ensure(g, 65);
  u32 bucket;
  const f64 u = int_float_pair8(rng_peek(g, g.pos + lane), bucket);
  const u32 layer = bucket >> 1;
  const f64 xi = c.lds.zig_norm_x[layer], xi1 = c.lds.zig_norm_x[layer + 1];
  const f64 x = u * xi;
  const bool fast = x < xi1;
  bool accept = fast, irregular = false;
  if (!fast) {
    if (layer == 0) {
      irregular = true;  // tail of the distribution
    } else {
      const f64 y01 = static_cast<f64>(rng_peek(g, g.pos + lane + 1)) * TWO_M64;
      if (!(y01 < 1.0)) {
        irregular = true;  // uniform_01 draws again
      } else {
        const f64 yi = c.lds.zig_norm_y[layer], yi1 = c.lds.zig_norm_y[layer + 1];
        const f64 y = yi + y01 * (yi1 - yi);
        const f64 chord = (xi - xi1) * y01 - (xi - x);
        const f64 tangent = y - (yi + (xi - x) * yi * xi);
        const f64 y_above_ubound = (xi >= 1) ? chord : tangent;
        const f64 y_above_lbound = (xi >= 1) ? tangent : chord;
        accept = y_above_ubound < 0 && (y_above_lbound < 0 || y < wave::f_exp(-(x * x / 2)));
      }
    }
  }
  const u32 mv = move_from_normal((bucket & 1u) ? x : -x, speed, std);
  // positions that start a two-output attempt: every other position of a run of slow positions
  u64 two = wave::ballot(!fast);
  u64 dbl = 0;
  while (two != 0) {
    const u32 b = static_cast<u32>(wave::ctz64(two));
    dbl |= u64(1) << b;
    two &= ~(u64(3) << b);
  }
  const u64 starts = ~(dbl << 1);
  const u64 irr = wave::ballot(irregular) & starts;
  const u32 stop = irr != 0 ? static_cast<u32>(wave::ctz64(irr)) : 64u;  // first irregular attempt
  const u64 below = stop < 64 ? lanemask_lt(stop) : ~u64(0);
  const u64 acc = wave::ballot(accept) & starts & below;
  wave::lockstep();  // queue slots read by the consumer of the previous step may be overwritten
  if ((acc >> lane) & 1u) {
    const u32 e = tail + static_cast<u32>(wave::popc64(acc & lanemask_lt(lane)));
    q_move[e % MOVQ_CAP] = mv;
    q_end[e % MOVQ_CAP] = static_cast<u32>(g.pos) + lane + 1 + static_cast<u32>((dbl >> lane) & 1u);
  }
  tail += static_cast<u32>(wave::popc64(acc));
  if (stop == 64) {
    rng_advance(g, 64 + static_cast<u32>(dbl >> 63));
  } else {
    rng_advance(g, stop);
    const f64 exact = unit_normal_exact(g, c.lds);
    if (lane == 0) {
      q_move[tail % MOVQ_CAP] = move_from_normal(exact, speed, std);
      q_end[tail % MOVQ_CAP] = static_cast<u32>(g.pos);
    }
    ++tail;
  }
  wave::sync_lds();
  return tail;
}

template <bool FWD>
MODLE_DEV_NOINLINE void generate_moves_dir(Cell& c, f64 speed, f64 std) {
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  u32* moves = FWD ? ws.f_move : ws.r_move;
  ensure_inverse<FWD>(c);
  const u32* rank = FWD ? ws.f_rank : ws.r_rank;
  if (std == 0.0) {
    const u32 move_int = static_cast<u32>(static_cast<u64>(wave::f_round(speed)));
    for (u32 base = 0; base < n; base += 64) {
      const u32 i = base + lane;
      if (i < n) moves[rank[i]] = ws.epoch[i] != UNBOUND ? move_int : 0;
    }
    return;
  }
  // Bound LEFs take the draws in id order.  The draws are produced 64 stream positions at a time
  // into a queue, independently of how the LEFs fall into batches; what the last step produced
  // beyond the draw of the last bound LEF is handed back by rewinding the stream position.
  const u32* q_move = c.lds.stage;
  const u32* q_end = c.lds.stage + MOVQ_CAP;
  u32 head = 0, tail = 0;  // entries consumed / produced
  constexpr u32 UX = 4;  // batches per group; the next group's loads go before this group's stores
  struct LefRegs {
    u32 E[UX], S[UX];
  };
  const auto load_lefs = [&](auto op, u32 group, LefRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 i = group + 64 * u + lane;
      r.E[u] = op(ws.epoch, i, i < n, UNBOUND, r.E[u]);
      r.S[u] = op(rank, i, i < n, 0, r.S[u]);
    }
  };
  LefRegs cur;
  load_lefs(wave::LdRaw{}, 0, cur);
  for (u32 group = 0; group < n; group += 64 * UX) {
    LefRegs g = cur;
    load_lefs(wave::LdMask{}, group, g);  // (defaults of the lanes outside the range)
    if (group + 64 * UX < n) load_lefs(wave::LdRaw{}, group + 64 * UX, cur);
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
    const u32 base = group + 64 * u;
    if (base >= n) break;
    const u32 i = base + lane;
    const bool act = i < n;
    const bool bnd = act && g.E[u] != UNBOUND;
    const u32 slot = g.S[u];
    const u64 bm = wave::ballot(bnd);
    const u32 need = static_cast<u32>(wave::popc64(bm));
    while (tail - head < need) tail = draw_moves_step(c, speed, std, tail);
    u32 mv = 0;
    if (bnd) mv = q_move[(head + static_cast<u32>(wave::popc64(bm & lanemask_lt(lane)))) % MOVQ_CAP];
    head += need;
    if (act) moves[slot] = mv;
    }
  }
  if (head != 0) {
    // the stream ends right after the draw of the last bound LEF: hand back what the last step
    // evaluated beyond it (queued draws, and rejected attempts that no draw followed)
    const u32 end_low = wave::uniform(q_end[(head - 1) % MOVQ_CAP]);
    c.g.pos = wave::known_uniform(c.g.pos - static_cast<u32>(static_cast<u32>(c.g.pos) - end_low));
  }
}

// The same when every active LEF is bound (always the case inside the epoch loop: generate_moves
// runs after select_and_bind_lefs): LEF i takes the i-th accepted draw of the direction, so the
// moves are a function of the stream alone.  They are stored in LEF-id order with coalesced
// stores -- no per-LEF state is read -- and the move adjustment, which walks the units in rank
// order, fetches each unit's move through its LEF id from this freshly written, compact array
// (instead of this pass scattering 4-byte stores over the rank-ordered array).  Measured on one
// box against the scattering form: 23 % fewer bytes written to the fabric and a kernel 0.7 %
// faster -- the adjustment pass itself takes twice as long (dependent gathers), every other pass
// gains from the lighter write traffic.
MODLE_DEV_NOINLINE void generate_moves_by_id(Cell& c, f64 speed, f64 std, u32* mv_by_id) {
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  if (std == 0.0) {
    const u32 move_int = static_cast<u32>(static_cast<u64>(wave::f_round(speed)));
    for (u32 base = 0; base < n; base += 64) {
      const u32 i = base + lane;
      if (i < n) mv_by_id[i] = move_int;
    }
    return;
  }
  const u32* q_move = c.lds.stage;
  const u32* q_end = c.lds.stage + MOVQ_CAP;
  u32 head = 0, tail = 0;  // entries consumed / produced
  for (u32 base = 0; base < n; base += 64) {
    const u32 need = umin(64u, n - base);
    while (tail - head < need) tail = draw_moves_step(c, speed, std, tail);
    if (lane < need) mv_by_id[base + lane] = q_move[(head + lane) % MOVQ_CAP];
    head += need;
  }
  if (head != 0) {
    // hand back what the last step evaluated beyond the draw of the last LEF
    const u32 end_low = wave::uniform(q_end[(head - 1) % MOVQ_CAP]);
    c.g.pos = wave::known_uniform(c.g.pos - static_cast<u32>(static_cast<u32>(c.g.pos) - end_low));
  }
}

// =============================================================================================
// adjust_moves_of_consecutive_extr_units (reference: simulation.cpp:350-407) as two segmented
// scans over rank order, fused with clamp_moves (reference: simulation.cpp:332-347).
//
// rev units, ranks high -> low:  land'[k] = min(land[k], land'[k+1] - 1) while both units are
// bound and neither reaches the 5'-end.  With d[k] = land[k] - k this is a segmented suffix
// minimum of d.  The reference tests "unit k+1 reaches the 5'-end" on the *updated* move of
// k+1; the scan uses the original move and the (rare, chromosome-end only) cases where the
// update changes the answer are replayed sequentially from the first affected rank.
// `do_adjust` / `do_clamp` exist for the phase-level test entry point.
// =============================================================================================
// The same two sweeps with FOUR consecutive ranks per lane (blocks of 256 ranks; used whenever the
// 32-bit scan applies, i.e. on every real chromosome): one 128-bit load per array and lane, the
// scan runs over the four units of a lane in registers, ONE cross-lane scan joins the 64 lanes,
// and the carries, the loop control and the violation test are paid once per 256 units instead
// of once per 64.  rev: lane 0 holds the highest ranks of a block and a lane walks its four units
// downwards, so that the suffix scan over ranks is again a prefix scan over (lane, unit).
// Returns the rank the sequential replay has to start from (adjust_moves_rev / _fwd), or -1.
template <bool FWD>
struct AdjustSweepX4 {
  struct Blk {
    wave::U32x4 P, M;
  };
  const u32 *pos, *uid, *mv_in, *mv_by_id;
  u32* mv_out;
  u32 n, lane, start, last, nblk;
  bool by_id, do_adjust, do_clamp;
  i32 carry_d;
  bool carry_ok, carry_cross;
  i64 viol_rank;
  wave::U32x4 ids;
  Blk cur;

  // first rank of this lane in block t of the sweep (t = 0 is the block the sweep starts with)
  MODLE_DEV_MEMBER u32 word0(u32 t) const { return (FWD ? t : nblk - 1 - t) * 256 + 4 * (FWD ? lane : 63 - lane); }
  // the ids of a block are requested one block ahead of its positions and (gathered) moves
  MODLE_DEV_MEMBER void load_ids(u32 t) {
    const u32 w = word0(t);
    ids = wave::ld4(uid, (by_id && t < nblk && w < n) ? w : 0u);
  }
  MODLE_DEV_MEMBER void load_blk(u32 t) {
    const u32 w = word0(t);
    const bool in = t < nblk && w < n;
    cur.P = wave::ld4(pos, in ? w : 0u);
    if (by_id) {
#pragma unroll
      for (u32 q = 0; q < 4; ++q) cur.M.v[q] = wave::LdRaw{}(mv_by_id, ids.v[q], in && w + q < n, 0, 0u);
    } else {
      cur.M = wave::ld4(mv_in, in ? w : 0u);
    }
  }
  MODLE_DEV_MEMBER void init(Cell& c, bool adjust, bool clamp, const u32* by_id_moves, u32* out) {
    Workspace& ws = c.ws;
    n = wave::uniform(c.n_active);
    lane = wave::lane();
    start = c.iv->start;
    last = c.iv->end - 1;
    pos = FWD ? ws.f_pos : ws.r_pos;
    uid = FWD ? ws.f_id : ws.r_id;
    mv_in = FWD ? ws.f_move : ws.r_move;
    mv_by_id = by_id_moves;
    mv_out = out;
    by_id = by_id_moves != nullptr;
    do_adjust = adjust;
    do_clamp = clamp;
    nblk = (n + 255) / 256;
    carry_d = 0;
    carry_ok = false;
    carry_cross = false;
    viol_rank = -1;
    load_ids(0);
    load_blk(0);
    if (1 < nblk) load_ids(1);
  }
  MODLE_DEV_MEMBER void step(u32 t) {
    const Blk g = cur;
    if (t + 1 < nblk) {
      load_blk(t + 1);
      if (t + 2 < nblk) load_ids(t + 2);
    }
    const u32 w = word0(t);
    u32 P[4], M[4], k[4];
    bool bnd[4], ok[4];
    i32 d[4];
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {  // j: position in sweep order inside the lane
      const u32 q = FWD ? j : 3 - j;
      k[j] = w + q;
      const bool act = k[j] < n;
      P[j] = g.P.v[q];
      M[j] = g.M.v[q];
      bnd[j] = act && P[j] != UNBOUND;
      if (FWD) {
        ok[j] = do_adjust && bnd[j] && static_cast<u64>(P[j]) + M[j] <= last;
        d[j] = ok[j] ? static_cast<i32>(P[j] + M[j] - k[j]) : 0;
      } else {
        ok[j] = do_adjust && bnd[j] && static_cast<u64>(P[j]) > static_cast<u64>(start) + M[j];
        d[j] = ok[j] ? static_cast<i32>(P[j] - M[j] - k[j]) : 0;
      }
    }
    const auto pick = [](i32 a, i32 b) { return FWD ? (a > b ? a : b) : (a < b ? a : b); };
    const bool ok_in = wave::shfl_up1(ok[3]);
    bool link[4], open[4];  // link: to the unit before; open: the chain reaches the start of the lane
    i32 v[4];
    link[0] = ok[0] && (lane > 0 ? ok_in : carry_ok);
    open[0] = link[0];
    v[0] = d[0];
#pragma unroll
    for (u32 j = 1; j < 4; ++j) {
      link[j] = ok[j] && ok[j - 1];
      v[j] = link[j] ? pick(d[j], v[j - 1]) : d[j];
      open[j] = link[j] && open[j - 1];
    }
    const SegScan inc = wave_prefix_segscan32<FWD>(v[3], open[3]);
    const i32 inc_val = static_cast<i32>(inc.val);
    const i32 whole = inc.cont ? pick(inc_val, carry_d) : inc_val;  // scan value of the lane's last unit
    const i32 whole_in = static_cast<i32>(wave::shfl_up1(static_cast<u32>(whole)));
    const i32 before = lane > 0 ? whole_in : carry_d;
    bool cross[4];
    wave::U32x4 O;
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      const i32 val = open[j] ? pick(v[j], before) : v[j];
      u32 Mnew = M[j];
      if (ok[j]) Mnew = FWD ? static_cast<u32>(val) + k[j] - P[j] : P[j] - (static_cast<u32>(val) + k[j]);
      cross[j] = ok[j] && (FWD ? static_cast<u64>(P[j]) + Mnew > last
                               : static_cast<u64>(P[j]) <= static_cast<u64>(start) + Mnew);
      O.v[FWD ? j : 3 - j] = (bnd[j] && do_clamp) ? umin(Mnew, FWD ? last - P[j] : P[j] - start) : Mnew;
    }
    if (w + 3 < n) {
      wave::st4(mv_out, w, O);
    } else {
#pragma unroll
      for (u32 q = 0; q < 4; ++q) {
        if (w + q < n) mv_out[w + q] = O.v[q];
      }
    }
    // first unit in sweep order whose link leads to a unit that crosses the end with its updated move
    const bool cross_in = wave::shfl_up1(cross[3]);
    const bool viol0 = link[0] && (lane > 0 ? cross_in : carry_cross);
    const bool viol1 = link[1] && cross[0], viol2 = link[2] && cross[1], viol3 = link[3] && cross[2];
    const u64 vm = wave::ballot(viol0 || viol1 || viol2 || viol3);
    if (vm != 0 && viol_rank < 0) {
      const u32 fl = static_cast<u32>(wave::ctz64(vm));
      const u32 jf = wave::bcast(viol0 ? 0u : viol1 ? 1u : viol2 ? 2u : 3u, fl);
      const u32 s = 4 * fl + jf;
      const u32 b = FWD ? t : nblk - 1 - t;
      viol_rank = FWD ? static_cast<i64>(b) * 256 + s - 1 : static_cast<i64>(b) * 256 + (255 - s) + 1;
    }
    carry_d = wave::bcast(whole, 63);
    carry_ok = wave::bcast(ok[3], 63);
    carry_cross = wave::bcast(cross[3], 63);
  }
};

template <bool FWD>
MODLE_DEV_NOINLINE i64 adjust_moves_x4(Cell& c, bool do_adjust, bool do_clamp, const u32* mv_by_id) {
  AdjustSweepX4<FWD> sw;
  sw.init(c, do_adjust, do_clamp, mv_by_id, c.ws.tmp[0]);
  for (u32 t = 0; t < sw.nblk; ++t) sw.step(t);
  return sw.viol_rank;
}

// `mv_by_id`: moves in LEF-id order (generate_moves_by_id) or nullptr when they already sit in
// r_move in rank order (phase-level test entry point).
// `out_slot` / `swept`: the scratch array the sweep writes (ws.tmp[out_slot]) and, when the sweep
// has been done already (adjust_moves_both_x4), the rank its replay starts from
MODLE_DEV_NOINLINE void adjust_moves_rev(Cell& c, bool do_adjust, bool do_clamp,
                                         const u32* mv_by_id = nullptr, u32 out_slot = 0,
                                         const i64* swept = nullptr) {
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  const u64 start = c.iv->start;
  // landing positions minus ranks fit 32 bits on every real chromosome: scans at half the cost
  const bool narrow = wave::uniform(c.iv->end) < 0x7F000000u;
  const u32* mv_in = ws.r_move;
  u32* mv_out = ws.tmp[out_slot];
  const u32 nbatch = (n + 63) / 64;
  const bool by_id = mv_by_id != nullptr;
  i64 carry_d = 0;
  bool carry_ok = false, carry_cross = false;
  i64 viol_rank = -1;
  // lanes hold the ranks of a batch in DESCENDING order (lane 0 = highest rank), so that the
  // suffix scan over ranks is a prefix scan over lanes
  if (swept != nullptr) {
    viol_rank = *swept;
  } else if (narrow) {
    viol_rank = adjust_moves_x4<false>(c, do_adjust, do_clamp, mv_by_id);
  } else {
  constexpr u32 UX = 4;  // batches per group; the next group's loads go before this group's stores
  struct UnitRegs {
    u32 P[UX], M[UX];
  };
  struct IdRegs {
    u32 I[UX];
  };
  // the ids of a group are requested one group ahead of its positions and (gathered) moves
  const auto load_ids = [&](u32 bg, IdRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const bool in = bg + u < nbatch;
      const u32 kq = (nbatch - 1 - (bg + u)) * 64 + (63 - lane);
      r.I[u] = wave::ld_sel(ws.r_id, kq, by_id && in && kq < n, 0);
    }
  };
  const auto load_units = [&](auto op, u32 bg, const IdRegs& ids, UnitRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const bool in = bg + u < nbatch;
      const u32 kq = (nbatch - 1 - (bg + u)) * 64 + (63 - lane);
      r.P[u] = op(ws.r_pos, kq, in && kq < n, UNBOUND, r.P[u]);
      r.M[u] = op(by_id ? mv_by_id : mv_in, by_id ? ids.I[u] : kq, in && kq < n, 0, r.M[u]);
    }
  };
  IdRegs ids;
  UnitRegs cur;
  load_ids(0, ids);
  load_units(wave::LdRaw{}, 0, ids, cur);
  if (UX < nbatch) load_ids(UX, ids);
  for (u32 bg = 0; bg < nbatch; bg += UX) {
    UnitRegs g = cur;
    load_units(wave::LdMask{}, bg, ids, g);  // (defaults of the lanes outside the range)
    if (bg + UX < nbatch) {
      load_units(wave::LdRaw{}, bg + UX, ids, cur);
      if (bg + 2 * UX < nbatch) load_ids(bg + 2 * UX, ids);
    }
    const u32* Pq = g.P;
    const u32* Mq = g.M;
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
    if (bg + u >= nbatch) break;
    const u32 bi = nbatch - 1 - (bg + u);
    const u32 k = bi * 64 + (63 - lane);
    const bool act = k < n;
    const u32 P = Pq[u];
    const u32 M = Mq[u];
    const bool bnd = act && P != UNBOUND;
    const bool okself = do_adjust && bnd && static_cast<u64>(P) > start + M;
    const i64 d = okself ? static_cast<i64>(P - M) - static_cast<i64>(k) : 0;
    const bool ok_next_in = wave::shfl_up1(okself);
    const bool ok_next = lane > 0 ? ok_next_in : carry_ok;
    const bool link = okself && ok_next;
    const SegScan sc = narrow ? wave_prefix_segscan32<false>(static_cast<i32>(d), link)
                              : wave_prefix_segscan<false>(SegScan{d, link});
    i64 val = sc.val;
    if (sc.cont) val = imin64(val, carry_d);
    u32 Mnew = M;
    if (okself) Mnew = P - static_cast<u32>(val + static_cast<i64>(k));
    const bool cross = okself && static_cast<u64>(P) <= start + Mnew;
    if (act) wave::st_stream(&mv_out[k], (bnd && do_clamp) ? umin(Mnew, P - static_cast<u32>(start)) : Mnew);
    const bool cross_next_in = wave::shfl_up1(cross);
    const bool cross_next = lane > 0 ? cross_next_in : carry_cross;
    const u64 vm = wave::ballot(link && cross_next);
    // highest rank k whose link to k+1 the scan got wrong (lowest lane); the replay starts at k+1
    if (vm != 0 && viol_rank < 0) viol_rank = bi * 64 + (63 - wave::ctz64(vm)) + 1;
    carry_d = wave::bcast(val, 63);
    carry_ok = wave::bcast(okself, 63);
    carry_cross = wave::bcast(cross, 63);
    }
  }
  }
  wave::sync_mem();
  if (viol_rank >= 0) {
    // sequential replay (reference loop) from the first rank whose decision the scan got wrong.
    // The reference adjusts all moves first and clamps afterwards, so the replay carries the
    // UNCLAMPED updated move of the unit it has just left (mv_out holds clamped values); the
    // unit it starts from is the one whose updated move crosses the 5'-end.
    bool first = true;
    u32 M2u = 0;
    for (u32 i = static_cast<u32>(viol_rank); i > 0; --i) {
      u32 M1 = by_id ? mv_by_id[ws.r_id[i - 1]] : mv_in[i - 1];
      const u64 P1 = ws.r_pos[i - 1], P2 = ws.r_pos[i];
      const bool both = P1 != UNBOUND && P2 != UNBOUND;
      if (both) {
        const bool cross2 = first || P2 <= start + M2u;
        if (!(P1 <= start + M1 || cross2)) {
          const u64 pos1 = P1 - M1, pos2 = P2 - M2u;
          if (pos2 <= pos1) M1 += static_cast<u32>(pos1 - pos2) + 1;
        }
      }
      M2u = M1;
      first = false;
      mv_out[i - 1] = (P1 != UNBOUND && do_clamp) ? umin(M1, static_cast<u32>(P1 - start)) : M1;
    }
    wave::sync_mem();
  }
  swap_ptr(ws.r_move, ws.tmp[out_slot]);
}

// `out_slot` / `swept`: the scratch array the sweep writes (ws.tmp[out_slot]) and, when the sweep
// has been done already (adjust_moves_both_x4), the rank its replay starts from
MODLE_DEV_NOINLINE void adjust_moves_fwd(Cell& c, bool do_adjust, bool do_clamp,
                                         const u32* mv_by_id = nullptr, u32 out_slot = 0,
                                         const i64* swept = nullptr) {
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  const u64 last = static_cast<u64>(c.iv->end) - 1;
  const bool narrow = wave::uniform(c.iv->end) < 0x7F000000u;  // see adjust_moves_rev
  const u32* mv_in = ws.f_move;
  u32* mv_out = ws.tmp[out_slot];
  const u32 nbatch = (n + 63) / 64;
  const bool by_id = mv_by_id != nullptr;
  i64 carry_d = 0;
  bool carry_ok = false, carry_cross = false;
  i64 viol_rank = -1;
  if (swept != nullptr) {
    viol_rank = *swept;
  } else if (narrow) {
    viol_rank = adjust_moves_x4<true>(c, do_adjust, do_clamp, mv_by_id);
  } else {
  constexpr u32 UX = 4;  // batches per group; the next group's loads go before this group's stores
  struct UnitRegs {
    u32 P[UX], M[UX];
  };
  struct IdRegs {
    u32 I[UX];
  };
  const auto load_ids = [&](u32 bg, IdRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 kq = (bg + u) * 64 + lane;
      r.I[u] = wave::ld_sel(ws.f_id, kq, by_id && kq < n, 0);
    }
  };
  const auto load_units = [&](auto op, u32 bg, const IdRegs& ids, UnitRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 kq = (bg + u) * 64 + lane;
      r.P[u] = op(ws.f_pos, kq, kq < n, UNBOUND, r.P[u]);
      r.M[u] = op(by_id ? mv_by_id : mv_in, by_id ? ids.I[u] : kq, kq < n, 0, r.M[u]);
    }
  };
  IdRegs ids;
  UnitRegs cur;
  load_ids(0, ids);
  load_units(wave::LdRaw{}, 0, ids, cur);
  if (UX < nbatch) load_ids(UX, ids);
  for (u32 bg = 0; bg < nbatch; bg += UX) {
    UnitRegs g = cur;
    load_units(wave::LdMask{}, bg, ids, g);  // (defaults of the lanes outside the range)
    if (bg + UX < nbatch) {
      load_units(wave::LdRaw{}, bg + UX, ids, cur);
      if (bg + 2 * UX < nbatch) load_ids(bg + 2 * UX, ids);
    }
    const u32* Pq = g.P;
    const u32* Mq = g.M;
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
    const u32 bi = bg + u;
    if (bi >= nbatch) break;
    const u32 k = bi * 64 + lane;
    const bool act = k < n;
    const u32 P = Pq[u];
    const u32 M = Mq[u];
    const bool bnd = act && P != UNBOUND;
    const bool okself = do_adjust && bnd && static_cast<u64>(P) + M <= last;
    const i64 d = okself ? static_cast<i64>(static_cast<u64>(P) + M) - static_cast<i64>(k) : 0;
    const bool ok_prev_in = wave::shfl_up1(okself);
    const bool ok_prev = lane > 0 ? ok_prev_in : carry_ok;
    const bool link = okself && ok_prev;  // link between k-1 and k
    const SegScan sc = narrow ? wave_prefix_segscan32<true>(static_cast<i32>(d), link)
                              : wave_prefix_segscan<true>(SegScan{d, link});
    i64 val = sc.val;
    if (sc.cont) val = imax64(val, carry_d);
    u32 Mnew = M;
    if (okself) Mnew = static_cast<u32>(val + static_cast<i64>(k) - static_cast<i64>(P));
    const bool cross = okself && static_cast<u64>(P) + Mnew > last;
    if (act) wave::st_stream(&mv_out[k], (bnd && do_clamp) ? umin(Mnew, static_cast<u32>(last - P)) : Mnew);
    const bool cross_prev_in = wave::shfl_up1(cross);
    const bool cross_prev = lane > 0 ? cross_prev_in : carry_cross;
    const u64 vm = wave::ballot(link && cross_prev);
    // lowest rank k-1 whose updated move crosses the 3'-end while the scan linked it to k
    if (vm != 0 && viol_rank < 0) viol_rank = static_cast<i64>(bi) * 64 + wave::ctz64(vm) - 1;
    carry_d = wave::bcast(val, 63);
    carry_ok = wave::bcast(okself, 63);
    carry_cross = wave::bcast(cross, 63);
    }
  }
  }
  wave::sync_mem();
  if (viol_rank >= 0) {
    // see adjust_moves_rev: the replay works on unclamped moves; unit viol_rank is the one whose
    // updated move crosses the 3'-end
    bool first = true;
    u32 M1u = 0;
    for (u32 i = static_cast<u32>(viol_rank) + 1; i < n; ++i) {
      u32 M2 = by_id ? mv_by_id[ws.f_id[i]] : mv_in[i];
      const u64 P1 = ws.f_pos[i - 1], P2 = ws.f_pos[i];
      const bool both = P1 != UNBOUND && P2 != UNBOUND;
      if (both) {
        const bool cross1 = first || P1 + M1u > last;
        if (!(cross1 || P2 + M2 > last)) {
          const u64 pos1 = P1 + M1u, pos2 = P2 + M2;
          if (pos1 >= pos2) M2 += static_cast<u32>(pos1 - pos2) + 1;
        }
      }
      M1u = M2;
      first = false;
      mv_out[i] = (P2 != UNBOUND && do_clamp) ? umin(M2, static_cast<u32>(last - P2)) : M2;
    }
    wave::sync_mem();
  }
  swap_ptr(ws.f_move, ws.tmp[out_slot]);
}

// Both sweeps in one loop: they are independent of each other (rev walks the blocks downwards, fwd
// upwards), so every iteration carries two dependency chains instead of one.
MODLE_DEV_NOINLINE void adjust_moves_both_x4(Cell& c, const u32* mv_rev, const u32* mv_fwd, i64& viol_rev,
                                             i64& viol_fwd) {
  AdjustSweepX4<false> r;
  AdjustSweepX4<true> f;
  r.init(c, true, true, mv_rev, c.ws.tmp[0]);
  f.init(c, true, true, mv_fwd, c.ws.tmp[1]);
  for (u32 t = 0; t < r.nblk; ++t) {
    r.step(t);
    f.step(t);
  }
  viol_rev = r.viol_rank;
  viol_fwd = f.viol_rank;
}

// `all_bound`: every active LEF is bound (the epoch loop's invariant at this point)
MODLE_DEV void phase_generate_moves(Cell& c, bool burnin_completed, bool all_bound = true) {
  const Params& p = *c.p;
  if (!all_bound) {
    PHASE(c, 5, generate_moves_dir<false>(c, burnin_completed ? p.rev_speed : p.rev_speed_burnin, p.rev_std);
          generate_moves_dir<true>(c, burnin_completed ? p.fwd_speed : p.fwd_speed_burnin, p.fwd_std);
          wave::sync_mem());
    PHASE(c, 6, adjust_moves_rev(c, true, true); adjust_moves_fwd(c, true, true));
    return;
  }
  // id-ordered moves live in scratch that is idle until the secondary pass lists its avoided
  // collisions there
  u32* mv_rev = c.ws.tmp[5];
  u32* mv_fwd = c.ws.tmp[6];
  PHASE(c, 5, generate_moves_by_id(c, burnin_completed ? p.rev_speed : p.rev_speed_burnin, p.rev_std, mv_rev);
        generate_moves_by_id(c, burnin_completed ? p.fwd_speed : p.fwd_speed_burnin, p.fwd_std, mv_fwd);
        wave::sync_mem());
  if (wave::uniform(c.iv->end) < 0x7F000000u) {  // (the 32-bit scans apply: see adjust_moves_rev)
    PHASE(c, 6, i64 vr; i64 vf; adjust_moves_both_x4(c, mv_rev, mv_fwd, vr, vf);
          wave::sync_mem();
          adjust_moves_rev(c, true, true, mv_rev, 0, &vr); adjust_moves_fwd(c, true, true, mv_fwd, 1, &vf));
  } else {
    PHASE(c, 6, adjust_moves_rev(c, true, true, mv_rev); adjust_moves_fwd(c, true, true, mv_fwd));
  }
}

// =============================================================================================
// ExtrusionBarriers::init_states / next_state (reference: extrusion_barriers.cpp:145-161,
// 219-230)
// =============================================================================================
MODLE_DEV_NOINLINE void barriers_init_states(Cell& c) {
  const Interval& iv = *c.iv;
  const u32 nb = wave::uniform(iv.n_barriers);
  const u32 lane = wave::lane();
  for (u32 base = 0; base < nb; base += 64) {
    const u32 i = base + lane;
    const bool act = i < nb;
    const f64 occ = wave::ld_sel(iv.bar_occupancy, i, act, 0.0);
    const bool draws = act && occ != 0.0;  // bernoulli(0) consumes nothing
    const u64 dm = wave::ballot(draws);
    const u32 cnt = static_cast<u32>(wave::popc64(dm));
    rng_ensure(c.g, cnt);
    const u32 k = static_cast<u32>(wave::popc64(dm & lanemask_lt(lane)));
    const bool on = draws && bernoulli_raw(rng_peek(c.g, c.g.pos + k), occ);
    if (act) c.ws.bar_active[i] = on ? 1 : 0;
    rng_advance(c.g, cnt);
  }
  wave::sync_mem();
}

// LEF-BAR detection without Bernoulli trials (both blocking probabilities in {0, 1}) works on
// the barriers that stall a unit, compacted in position order: list 0 as the rev units see
// them, list 1 as the fwd units do.  A barrier is on a list iff it is active and the blocking
// probability that applies to it there is 1.
MODLE_DEV bool stalling_lists_wanted(const Params& p) {
  return (p.pblock_major == 1.0 || p.pblock_major == 0.0) &&
         (p.pblock_minor == 1.0 || p.pblock_minor == 0.0);
}
constexpr u32 HITBAR_HARD = 0x80000000u;

// appends the barriers of one batch (index i per lane, `on`: active) to the two lists; uniform
MODLE_DEV void stalling_lists_append(Cell& c, u32 i, bool in, bool on, u32 bpos, u32 bdir) {
  const Params& p = *c.p;
  const u32 lane = wave::lane();
#pragma unroll
  for (u32 d = 0; d < 2; ++d) {
    const bool is_major = bdir == (d == 0 ? DIR_REV : DIR_FWD);
    const bool hit = in && on && ((is_major ? p.pblock_major : p.pblock_minor) == 1.0);
    const u64 hm = wave::ballot(hit);
    if (hit) {
      const u32 slot = c.n_hit[d] + static_cast<u32>(wave::popc64(hm & lanemask_lt(lane)));
      c.ws.hit_pos[d][slot] = bpos;
      c.ws.hit_idx[d][slot] = i | (is_major ? HITBAR_HARD : 0u);
    }
    c.n_hit[d] += static_cast<u32>(wave::popc64(hm));
  }
}

// stand-alone construction of the lists from the current barrier states (phase-level test entry
// point; the epoch loop builds them while it updates the states)
MODLE_DEV_NOINLINE void compact_stalling_barriers(Cell& c) {
  const Interval& iv = *c.iv;
  const u32 nb = wave::uniform(iv.n_barriers);
  const u32 lane = wave::lane();
  c.n_hit[0] = 0;
  c.n_hit[1] = 0;
  for (u32 base = 0; base < nb; base += 64) {
    const u32 i = base + lane;
    const bool in = i < nb;
    stalling_lists_append(c, i, in, in && c.ws.bar_active[i] != 0, in ? iv.bar_pos[i] : 0,
                          in ? iv.bar_dir[i] : 0);
  }
  wave::sync_mem();
}

MODLE_DEV_NOINLINE void barriers_next_state(Cell& c) {
  const Interval& iv = *c.iv;
  const u32 nb = wave::uniform(iv.n_barriers);
  const u32 lane = wave::lane();
  const bool lists = stalling_lists_wanted(*c.p);
  c.n_hit[0] = 0;
  c.n_hit[1] = 0;
  constexpr u32 UX = 4;  // batches per group; the next group's loads go before this group's stores
  struct BarRegs {
    u8 S[UX], D[UX];
    u32 P[UX];
    f64 I[UX], A[UX];
  };
  const auto load_bars = [&](auto op, u32 group, BarRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 iq = group + 64 * u + lane;
      r.S[u] = op(c.ws.bar_active, iq, iq < nb, u8(0), r.S[u]);
      r.I[u] = op(iv.bar_stp_inactive, iq, iq < nb, 0.0, r.I[u]);
      r.A[u] = op(iv.bar_stp_active, iq, iq < nb, 0.0, r.A[u]);
      r.D[u] = op(iv.bar_dir, iq, lists && iq < nb, u8(0), r.D[u]);
      r.P[u] = op(iv.bar_pos, iq, lists && iq < nb, 0, r.P[u]);
    }
  };
  BarRegs cur{};
  if (nb != 0) load_bars(wave::LdRaw{}, 0, cur);
  for (u32 group = 0; group < nb; group += 64 * UX) {
    BarRegs g = cur;
    load_bars(wave::LdMask{}, group, g);  // (defaults of the lanes outside the range)
    if (group + 64 * UX < nb) load_bars(wave::LdRaw{}, group + 64 * UX, cur);
    const u8* Sq = g.S;
    const f64* Iq = g.I;
    const f64* Aq = g.A;
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 base = group + 64 * u;
      if (base >= nb) break;
      const u32 i = base + lane;
      const u32 cnt = umin(64u, nb - base);
      rng_ensure(c.g, cnt);
      u8 st = Sq[u];
      if (i < nb) {
        const f64 r = canonical_raw(rng_peek(c.g, c.g.pos + lane));
        if (!st && r > Iq[u]) {
          st = 1;
          c.ws.bar_active[i] = 1;
        } else if (st && r > Aq[u]) {
          st = 0;
          c.ws.bar_active[i] = 0;
        }
      }
      rng_advance(c.g, cnt);
      if (lists) stalling_lists_append(c, i, i < nb, st != 0, g.P[u], g.D[u]);
    }
  }
  wave::sync_mem();
}

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
// The barriers a batch of 64 consecutive ranks can touch form one index range that continues
// where the previous batch stopped.  A window of BAR_WIN barriers (position and a flag word:
// state, blocking direction) is staged in LDS with one coalesced load and all per-unit searches
// run there; a batch whose units need more than the window falls back to device memory.
// The window lives in the LDS sort buffer (idle during the collision passes): BAR_WIN positions
// followed by BAR_WIN flag words.  It is re-staged only when a batch starts closer than BAR_NEED
// barriers to its far edge.
constexpr u32 BAR_WIN = SORT_LDS_CAP;  // SORT_LDS_CAP u64 keys = 2 * BAR_WIN words
constexpr u32 BAR_NEED = 128;

// Position of the barrier that stalls the unit of rank k (valid where the collision word says
// LEF-BAR), written by detect_lef_bar for the passes that correct moves.  Lives in ranking
// scratch, which is idle during the collision passes.
template <bool FWD>
MODLE_DEV u32* stalling_barrier_positions(const Workspace& ws) {
  return FWD ? ws.tmp[4] : ws.tmp[3];
}

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

// Bernoulli trials of one unit: how many it consumes (count_only) or which barrier stalls it
template <bool FWD, bool STAGED_ONLY>
MODLE_DEV u32 lef_bar_count_trials(const BarView& v, const Params& p, u32 b_lo, u32 b_hi) {
  const u32 major_dir = FWD ? DIR_FWD : DIR_REV;
  u32 ntr = 0;
  for (u32 b = b_lo; b < b_hi; ++b) {
    const u32 fl = v.flag<STAGED_ONLY>(b);
    const f64 pb = (fl >> 1) == major_dir ? p.pblock_major : p.pblock_minor;
    ntr += ((fl & 1u) && pb != 1.0 && pb != 0.0) ? 1u : 0u;
  }
  return ntr;
}

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

// ---------------------------------------------------------------------------------------------
// detect_lef_bar_collisions when both blocking probabilities are 0 or 1 (the reference default:
// major 1, minor 0): no Bernoulli trial is drawn and a barrier stalls a unit iff it is active
// and the probability that applies to its direction is 1.  Of the barriers in a unit's window the
// reference keeps the one it visits last: the highest such barrier for a rev unit, the lowest
// for a fwd unit.  The stalling barriers are therefore compacted (position, index | hard << 31)
// into the LDS window, in ascending order, and a unit needs one search there and one test.
//
// The compacted window holds every stalling barrier with index in [s0, s1); it serves any unit
// window [lo, hi) with lo >= lo_cover and hi <= hi_cover.  Unit windows are disjoint and ordered
// like the ranks, so every batch continues the search where the previous one stopped; a batch
// whose windows do not fit is looked up in device memory.
// ---------------------------------------------------------------------------------------------
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
  const u32* moves = FWD ? ws.f_move : ws.r_move;
  u32* coll = FWD ? ws.f_coll : ws.r_coll;
  u32* barpos = stalling_barrier_positions<FWD>(ws);
  u32* cp = reinterpret_cast<u32*>(c.lds.sort_lds);  // positions of the compacted barriers
  u32* ci = cp + BAR_WIN;                              // their indices (| HITBAR_HARD)
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
      lo_key[j] = 0;
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
          g1 = umin(g0 + BAR_WIN, nh);
        } else {
          g0 = g1 > BAR_WIN ? g1 - BAR_WIN : 0;
        }
        cnt = g1 - g0;
        wave::lockstep();
        {
          // what the window does not hold: everything before it lies below lo_cover,
          // everything after it at or above hi_cover
          const u32 edge_lo = g0 > 0 ? hpos[g0 - 1] : 0;
          const u32 edge_hi = g1 < nh ? hpos[g1] : 0;
          stage_stalling_window_call((MODLE_LDS u32*)cp, (MODLE_LDS u32*)ci, hpos + g0, hidx + g0, cnt);
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
          bool in[4];
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            jx[j] = q[j] + sft;
            in[j] = bnd[j] & (jx[j] <= cnt);
            kv[j] = cp[c0 + (in[j] ? jx[j] - 1 : 0)];  // (no branch around the read)
          }
          wave::sched_fence();
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            if (in[j] & (kv[j] < hi_key[j])) q[j] = jx[j];
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
          bool in[4];
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            in[j] = bnd[j] & (q[j] >= sft);
            kv[j] = cp[c0 + (in[j] ? q[j] - sft : 0)];  // (no branch around the read)
          }
          wave::sched_fence();
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            if (in[j] & (kv[j] >= lo_key[j])) q[j] -= sft;
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

template <bool FWD>
MODLE_DEV_NOINLINE void detect_lef_bar(Cell& c, BoundaryCounts bc) {
  Workspace& ws = c.ws;
  const Interval& iv = *c.iv;
  const Params& p = *c.p;
  const u32 n = wave::uniform(c.n_active);
  const u32 nb = wave::uniform(iv.n_barriers);
  if (nb == 0) return;
  const u32 lane = wave::lane();
  const u32* pos = FWD ? ws.f_pos : ws.r_pos;
  const u32* moves = FWD ? ws.f_move : ws.r_move;
  u32* coll = FWD ? ws.f_coll : ws.r_coll;
  const bool trials = !((p.pblock_major == 1.0 || p.pblock_major == 0.0) &&
                        (p.pblock_minor == 1.0 || p.pblock_minor == 0.0));
  if (!trials) {
    detect_lef_bar_det<FWD>(c, bc);
    return;
  }
  u32* barpos = stalling_barrier_positions<FWD>(ws);
  u32* st_pos = reinterpret_cast<u32*>(c.lds.sort_lds);
  u32* st_flag = st_pos + BAR_WIN;
  // first / last rank that takes part
  const u32 j_rev0 = bc.n5 == 0 ? 0 : bc.n5 - 1;
  const u32 j_fwd0 = bc.n3 == 0 ? n - 1 : n - bc.n3;
  u32 carry_pos = 0;  // position of the neighbouring unit processed by the previous batch
  u32 anchor = 0;     // rev: first barrier index the next batch can need; fwd: one past the last
  BarView v;
  v.iv = &iv;
  v.active = ws.bar_active;
  v.st_pos = st_pos;
  v.st_flag = st_flag;
  v.s0 = 0;
  v.s1 = 0;  // nothing staged yet
  bool located = false;
  const u32 nbatch = (n + 63) / 64;
  constexpr u32 UX = 4;  // batches whose loads are in flight together
  for (u32 bg = 0; bg < nbatch; bg += UX) {
    u32 Pq[UX], Mq[UX];
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      // rev: ranks ascending; fwd: ranks descending, lane 0 = highest rank of the batch
      const i64 kk = FWD ? static_cast<i64>(j_fwd0) - static_cast<i64>(bg + u) * 64 - lane
                         : static_cast<i64>(j_rev0) + static_cast<i64>(bg + u) * 64 + lane;
      const bool act = kk >= 0 && kk < static_cast<i64>(n);
      Pq[u] = wave::ld_sel(pos, static_cast<u32>(kk), act, 0);
      Mq[u] = wave::ld_sel(moves, static_cast<u32>(kk), act, 0);
    }
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 bi = bg + u;
      const i64 kk = FWD ? static_cast<i64>(j_fwd0) - static_cast<i64>(bi) * 64 - lane
                         : static_cast<i64>(j_rev0) + static_cast<i64>(bi) * 64 + lane;
      const bool act = kk >= 0 && kk < static_cast<i64>(n);
      if (!wave::any(act)) break;
      const u32 k = act ? static_cast<u32>(kk) : 0;
      const u32 P = Pq[u];
      const u32 M = Mq[u];
      const bool bnd = act && P != UNBOUND;
      // neighbour towards which the barriers are shadowed (rank k-1 for rev, k+1 for fwd)
      const u32 nbr_in = wave::shfl_up1(P);
      const bool first = (bi == 0 && lane == 0);
      const u32 nbr = lane > 0 ? nbr_in : carry_pos;
      // the unit can be stalled by barriers with lo_key <= position < hi_key
      u64 lo_key = 0, hi_key = 0;
      if (bnd) {
        if (!FWD) {
          // prev <= bpos < P and P - bpos <= M
          const u32 reach = P - M;  // M <= P - start after clamping
          lo_key = first ? reach : umax(reach, nbr);
          hi_key = P;
        } else {
          // P < bpos <= next and bpos - P <= M
          const u64 reach = static_cast<u64>(P) + M;
          lo_key = static_cast<u64>(P) + 1;
          hi_key = (first ? reach : umin64(reach, nbr)) + 1;
        }
      }
      const u64 bm = wave::ballot(bnd);
      carry_pos = wave::bcast(P, 63);
      if (bm == 0) continue;
      // the first batch with a bound unit locates the window through the bucket table; later
      // batches continue where the previous one stopped
      if (!located) {
        const u32 l0 = static_cast<u32>(wave::ctz64(bm));
        const u64 key = FWD ? wave::bcast(hi_key, l0) : wave::bcast(lo_key, l0);
        anchor = wave::uniform(bar_lower_bound(iv, key));
        located = true;
      }
      const bool restage = FWD ? (v.s1 == 0 || (anchor < v.s0 + BAR_NEED && v.s0 > 0) || anchor > v.s1)
                               : (v.s1 == 0 || (anchor + BAR_NEED > v.s1 && v.s1 < nb) || anchor < v.s0);
      if (restage) {
        if (!FWD) {
          v.s0 = anchor;
          v.s1 = umin(anchor + BAR_WIN, nb);
        } else {
          v.s1 = anchor;
          v.s0 = anchor > BAR_WIN ? anchor - BAR_WIN : 0;
        }
        wave::lockstep();
        for (u32 t = lane; t < BAR_WIN; t += 64) {
          const u32 b = v.s0 + t;
          if (b < v.s1) {
            st_pos[t] = iv.bar_pos[b];
            st_flag[t] =
                static_cast<u32>(ws.bar_active[b] != 0) | (static_cast<u32>(iv.bar_dir[b]) << 1);
          }
        }
        wave::sync_lds();
      }
      // windows of barrier indices [b_lo, b_hi): in LDS when every unit of the batch stays inside
      // the staged range, otherwise through the general accessors
      u32 b_lo = 0, b_hi = 0;
      bool edge = false;
      if (bnd) lef_bar_window<FWD, true>(v, nb, anchor, lo_key, hi_key, b_lo, b_hi, edge);
      const bool staged_only = !wave::any(edge);
      if (!staged_only) {
        b_lo = 0;
        b_hi = 0;
        if (bnd) lef_bar_window<FWD, false>(v, nb, anchor, lo_key, hi_key, b_lo, b_hi, edge);
      }
      // number of Bernoulli trials this unit consumes
      u32 ntr = 0;
      if (trials) {
        ntr = staged_only ? lef_bar_count_trials<FWD, true>(v, p, b_lo, b_hi)
                          : lef_bar_count_trials<FWD, false>(v, p, b_lo, b_hi);
      }
      u32 off = 0, total = 0;
      if (trials) {
        // exclusive prefix sum of ntr over lanes
        off = wave_prefix_sum_u32(ntr);
        total = wave::bcast(off, 63);
        off -= ntr;
      }
      bool hard = false;
      u32 bpos = 0;
      u32 winner = 0xFFFFFFFFu;
      if (total <= RNG_BLOCK) {
        if (total != 0) rng_ensure(c.g, total);
        winner = staged_only ? lef_bar_pick<FWD, true>(v, p, c.g, b_lo, b_hi, off, hard, bpos)
                             : lef_bar_pick<FWD, false>(v, p, c.g, b_lo, b_hi, off, hard, bpos);
        rng_advance(c.g, total);
      } else {
        // More Bernoulli trials in this batch than one block of the PRNG ring serves (dense
        // barrier annotations with a fractional blocking probability): the lanes are resolved in
        // rounds, each taking the longest run of lanes (in lane = stream order) whose trials fit
        // one block; a single unit with more trials than that is replayed sequentially.
        u64 pend = wave::ballot(bnd);
        u32 base_tr = 0;  // trials consumed by the lanes resolved so far
        while (pend != 0) {
          const bool mine_pending = ((pend >> lane) & 1u) != 0;
          const bool fits = mine_pending && (off + ntr - base_tr <= RNG_BLOCK);
          const u64 fm = wave::ballot(fits);
          if (fm == 0) {
            const u32 l = static_cast<u32>(wave::ctz64(pend));
            const u32 lo = wave::bcast(b_lo, l), hi = wave::bcast(b_hi, l);
            const u32 major_dir = FWD ? DIR_FWD : DIR_REV;
            u32 w = 0xFFFFFFFFu;
            bool h = false;
            for (u32 q = lo; q < hi; ++q) {
              const u32 b = FWD ? (hi - 1 - (q - lo)) : q;  // reference visiting order
              const u32 fl = wave::uniform(v.flag<false>(b));
              if (!(fl & 1u)) continue;
              const f64 pb = (fl >> 1) == major_dir ? p.pblock_major : p.pblock_minor;
              bool hit;
              if (pb == 1.0) {
                hit = true;
              } else if (pb == 0.0) {
                hit = false;
              } else {
                hit = bernoulli_raw(rng_next(c.g), pb);
              }
              if (hit) {
                w = b;
                h = (fl >> 1) == major_dir;
              }
            }
            if (lane == l) {
              winner = w;
              hard = h;
              if (w != 0xFFFFFFFFu) bpos = v.pos<false>(w);
            }
            base_tr += wave::bcast(ntr, l);
            pend &= ~(u64(1) << l);
          } else {
            // fitting lanes are a run of pending lanes starting at the first one
            const u32 l_last_fit = static_cast<u32>(63 - wave::clz64(fm));
            const u32 cnt = wave::bcast(off + ntr, l_last_fit) - base_tr;
            if (cnt != 0) rng_ensure(c.g, cnt);
            if (fits) {
              winner = staged_only
                           ? lef_bar_pick<FWD, true>(v, p, c.g, b_lo, b_hi, off - base_tr, hard, bpos)
                           : lef_bar_pick<FWD, false>(v, p, c.g, b_lo, b_hi, off - base_tr, hard, bpos);
            }
            rng_advance(c.g, cnt);
            base_tr += cnt;
            pend &= ~fm;
          }
        }
      }
      if (winner != 0xFFFFFFFFu) {
        coll[k] = cw_make(winner, EV_COLLISION | EV_LEF_BAR) | (hard ? CW_HARD : 0u);
        barpos[k] = bpos;
      }
      // where the next batch continues: past the last bound unit's window (rev) / below it (fwd)
      const u32 l_last = static_cast<u32>(63 - wave::clz64(bm));
      anchor = FWD ? wave::bcast(b_lo, l_last) : wave::bcast(b_hi, l_last);
    }
  }
  wave::sync_mem();
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
// what one batch of detect_primary reads from device memory: the rev units of 64 ranks and a
// slice of STAGE_CAP fwd units
// (positions for the whole slice: every lane searches them; moves, collision words and ids for
// its first PRIMARY_NEAR units only: a rev unit's partner is almost always among them)
constexpr u32 PRIMARY_NEAR = 256;
// one block of detect_primary: TWO consecutive rev ranks per lane (128 ranks, 64-bit loads) and
// the slices of the fwd-side arrays
struct PrimaryBatch {
  wave::U32x2 R, rev_move, rev_id, rc, rbp;
  u32 sp[STAGE_CAP / 64];
  u32 sm[PRIMARY_NEAR / 64], sc[PRIMARY_NEAR / 64], si[PRIMARY_NEAR / 64], sb[PRIMARY_NEAR / 64];
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
    b.rev_id = wave::ld2(ws.r_id, kq);
    b.rc = wave::ld2(ws.r_coll, kq);
    // position of the barrier that stalls the unit (meaningful where the word says LEF-BAR):
    // having it here keeps a dependent load, and with it a wait for everything in flight, out of
    // the block's work
    b.rbp = wave::ld2(stalling_barrier_positions<false>(ws), kq);
  }
#pragma unroll
  for (u32 q = 0; q < STAGE_CAP / 64; ++q) {
    const u32 t = lane + 64 * q;
    b.sp[q] = op(ws.f_pos, w0 + t, w0 + t < n, UNBOUND, b.sp[q]);
  }
#pragma unroll
  for (u32 q = 0; q < PRIMARY_NEAR / 64; ++q) {
    const u32 t = lane + 64 * q;
    const bool in = w0 + t < n;
    b.sm[q] = op(ws.f_move, w0 + t, in, 0, b.sm[q]);
    b.sc[q] = op(ws.f_coll, w0 + t, in, 0, b.sc[q]);
    b.si[q] = op(ws.f_id, w0 + t, in, 0, b.si[q]);
    b.sb[q] = op(stalling_barrier_positions<true>(ws), w0 + t, in, 0, b.sb[q]);
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
  // LDS slices of the fwd-side arrays, ranks [w0, w0 + STAGE_CAP): positions in the staging
  // buffer, moves / collision words / ids / barrier positions in the (idle) sort buffer
  u32* stage = c.lds.stage;
  u32* st_move = reinterpret_cast<u32*>(c.lds.sort_lds);
  u32* st_coll = st_move + PRIMARY_NEAR;
  u32* st_id = st_coll + PRIMARY_NEAR;
  u32* st_bp = st_id + PRIMARY_NEAR;
  static_assert(4 * PRIMARY_NEAR <= 2 * SORT_LDS_CAP, "fwd slices do not fit the sort buffer");
  u32 carry_pos = 0;
  u32 carry_pf = 0;  // fwd units strictly upstream of the last rev unit handled so far
  // pf = number of fwd units strictly upstream of R.  pf is monotone in the rank, so slices of
  // the fwd arrays starting at the previous block's value are staged in LDS (one round trip
  // together with the block's rev-side loads) and everything is looked up there; units whose
  // partner lies beyond the slice use device memory.  The loads of the next block are issued as
  // soon as this block knows where its last unit falls among the fwd units, before the rest of
  // its work.  What they can miss are this block's updates of the fwd unit at the start of the
  // next slice (its move and collision word), and no unit of the next block can pair with that
  // unit: it lies upstream of this block's last rev unit, which is then the "first rev unit
  // downstream of it".
  const u32 first = bc.n5;
  PrimaryBatch cur;
  primary_load_batch(wave::LdRaw{}, ws, n, first & ~1u, 0, lane, cur, true);
  for (u32 base = first & ~1u; base < n; base += 128) {
    const u32 w0 = carry_pf > 0 ? carry_pf - 1 : 0;
    primary_load_batch(wave::LdMask{}, ws, n, base, w0, lane, cur, false);  // (defaults outside the range)
    u32 k[2], R[2], rev_move_k[2], rev_id_k[2], rc_k[2], rbp_k[2];
    bool act[2];
#pragma unroll
    for (u32 j = 0; j < 2; ++j) {
      k[j] = base + 2 * lane + j;
      act[j] = k[j] >= first && k[j] < n;
      R[j] = act[j] ? cur.R.v[j] : UNBOUND;
      rev_move_k[j] = act[j] ? cur.rev_move.v[j] : 0u;
      rev_id_k[j] = act[j] ? cur.rev_id.v[j] : 0u;
      rc_k[j] = act[j] ? cur.rc.v[j] : 0u;
      rbp_k[j] = act[j] ? cur.rbp.v[j] : 0u;
    }
    wave::lockstep();
#pragma unroll
    for (u32 q = 0; q < STAGE_CAP / 64; ++q) stage[lane + 64 * q] = cur.sp[q];
#pragma unroll
    for (u32 q = 0; q < PRIMARY_NEAR / 64; ++q) {
      const u32 t = lane + 64 * q;
      st_move[t] = cur.sm[q];
      st_coll[t] = cur.sc[q];
      st_id[t] = cur.si[q];
      st_bp[t] = cur.sb[q];
    }
    wave::sync_lds();
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
#pragma unroll
    for (u32 j = 0; j < 2; ++j) {
      if (act[j]) {
        u32 l = lo[j];
        if (l == STAGE_CAP - 1 && st_last < R[j]) l = STAGE_CAP;
        if (l == STAGE_CAP && w0 + STAGE_CAP < n) {
          pf[j] = lower_bound_u32(ws.f_pos, n, R[j]);
        } else {
          pf[j] = umin(w0 + l, n);
        }
      }
    }
    // pf of the last active unit
    const u64 am = wave::ballot(act[0] || act[1]);
    const u32 next_pf = wave::bcast(act[1] ? pf[1] : pf[0], static_cast<u32>(63 - wave::clz64(am)));
    if (base + 128 < n) {
      primary_load_batch(wave::LdRaw{}, ws, n, base + 128, next_pf > 0 ? next_pf - 1 : 0, lane, cur, true);
    }
    // the partner of each unit (the fwd unit right upstream of it): its five words are read from
    // the slices together, without branches; partners beyond the slices come from device memory
    static_assert(PRIMARY_NEAR == STAGE_CAP, "one staged range for all five fwd-side arrays");
    bool cand[2] = {false, false};
    u32 F[2], rev_move[2], fwd_move[2], fwd_id_s[2], fc_s[2], fbp_s[2];
    bool has[2], staged[2];
#pragma unroll
    for (u32 j = 0; j < 2; ++j) {
      has[j] = act[j] && pf[j] >= 1 && pf[j] < i2;
      const u32 kf = pf[j] - 1;
      staged[j] = has[j] && kf >= w0 && kf - w0 < STAGE_CAP;
      const u32 e = staged[j] ? kf - w0 : 0u;
      F[j] = stage[e];
      fwd_move[j] = st_move[e];
      fwd_id_s[j] = st_id[e];
      fc_s[j] = st_coll[e];
      fbp_s[j] = st_bp[e];
      rev_move[j] = rev_move_k[j];
    }
    wave::sched_fence();
    if (wave::any((has[0] && !staged[0]) || (has[1] && !staged[1]))) {
#pragma unroll
      for (u32 j = 0; j < 2; ++j) {
        if (has[j] && !staged[j]) {
          const u32 kf = pf[j] - 1;
          F[j] = ws.f_pos[kf];
          fwd_move[j] = ws.f_move[kf];
          fwd_id_s[j] = ws.f_id[kf];
          fc_s[j] = ws.f_coll[kf];
          fbp_s[j] = stalling_barrier_positions<true>(ws)[kf];
        }
        // (the loads end HERE: where values loaded on a rare path merge with the common path the
        // compiler waits for everything in flight -- the next block's loads -- on both)
        wave::pin(F[j]);
        wave::pin(fwd_move[j]);
        wave::pin(fwd_id_s[j]);
        wave::pin(fc_s[j]);
        wave::pin(fbp_s[j]);
      }
    }
#pragma unroll
    for (u32 j = 0; j < 2; ++j) {
      const u32 Rprev = j == 0 ? Rprev0 : R[0];
      const bool first_after = (k[j] == first) || Rprev <= F[j];
      const u32 delta = R[j] - F[j];  // > 0 by construction (where it is used)
      cand[j] = has[j] && first_after && static_cast<u64>(delta) < static_cast<u64>(rev_move[j]) + fwd_move[j];
    }
    const u64 cm0 = wave::ballot(cand[0]), cm1 = wave::ballot(cand[1]);
    bool hit[2] = {cand[0] && !never_collide, cand[1] && !never_collide};
    if (trials && (cm0 | cm1) != 0) {
      const u32 cnt = static_cast<u32>(wave::popc64(cm0) + wave::popc64(cm1));
      rng_ensure(c.g, cnt);
      // draws in rank order: unit (lane, j) after the units of the lanes before it and after unit 0
      // of its own lane
      const u64 lt = lanemask_lt(lane);
      const u32 t0 = static_cast<u32>(wave::popc64(cm0 & lt) + wave::popc64(cm1 & lt));
      const u32 t1 = t0 + (cand[0] ? 1u : 0u);
      hit[0] = cand[0] && bernoulli_raw(rng_peek(c.g, c.g.pos + t0), p_collide);
      hit[1] = cand[1] && bernoulli_raw(rng_peek(c.g, c.g.pos + t1), p_collide);
      rng_advance(c.g, cnt);
    }
    // The collisions are rare (a few units per block) and their handling is long divergent code:
    // it runs once for the lane's unit that collided, and a second time only when both units of a
    // lane did.
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
    if (wave::any(hit[0] || hit[1])) {
      const u32 h = hit[0] ? 0u : 1u;
      if (hit[0] || hit[1]) {
        handle_hit(h ? pf[1] : pf[0], h ? k[1] : k[0], h ? R[1] : R[0], h ? F[1] : F[0], h ? rev_move[1] : rev_move[0],
                   h ? fwd_move[1] : fwd_move[0], h ? rev_id_k[1] : rev_id_k[0], h ? fwd_id_s[1] : fwd_id_s[0],
                   h ? rc_k[1] : rc_k[0], h ? fc_s[1] : fc_s[0], h ? rbp_k[1] : rbp_k[0], h ? fbp_s[1] : fbp_s[0]);
      }
      if (wave::any(hit[0] && hit[1])) {
        if (hit[0] && hit[1]) {
          handle_hit(pf[1], k[1], R[1], F[1], rev_move[1], fwd_move[1], rev_id_k[1], fwd_id_s[1], rc_k[1], fc_s[1],
                     rbp_k[1], fbp_s[1]);
        }
      }
    }
    carry_pos = wave::bcast(R[1], 63);
    carry_pf = next_pf;
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
template <bool FWD>
struct SecondaryFilter {
  static constexpr u32 UX = 4;  // batches per group; the next group's loads go before this group's stores
  struct UnitRegs {
    u32 P[UX], M[UX], C[UX], B[UX];
  };
  const u32 *pos, *coll, *barpos;
  u32 *moves, *q_k, *dump;
  u32 n, lane, nbatch, cap, n_cand, carry_pos, carry_coll;
  i32 f_first;
  bool correct_lef_bar, do_secondary, carry_pending;
  UnitRegs cur;

  MODLE_DEV_MEMBER void load_units(bool raw, u32 bg, UnitRegs& r) const {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 bi = bg + u;
      // (ranks stay far below 2^31: 32-bit index arithmetic)
      const i32 kk = FWD ? static_cast<i32>(n) - 1 - static_cast<i32>(bi * 64 + lane)
                         : static_cast<i32>(bi * 64 + lane);
      const bool act = kk >= 0 && static_cast<u32>(kk) < n;
      const u32 k = act ? static_cast<u32>(kk) : 0;
      if (raw) {
        r.P[u] = wave::LdRaw{}(pos, k, act, 0, r.P[u]);
        r.M[u] = wave::LdRaw{}(moves, k, act, 0, r.M[u]);
        r.C[u] = wave::LdRaw{}(coll, k, act, 0, r.C[u]);
        r.B[u] = wave::LdRaw{}(barpos, k, act, 0, r.B[u]);
      } else {
        r.P[u] = wave::LdMask{}(pos, k, act, 0, r.P[u]);
        r.M[u] = wave::LdMask{}(moves, k, act, 0, r.M[u]);
        r.C[u] = wave::LdMask{}(coll, k, act, 0, r.C[u]);
        r.B[u] = wave::LdMask{}(barpos, k, act, 0, r.B[u]);
      }
    }
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
    // cannot be counted by the compiler, and the wait for the next group's loads then becomes a
    // wait for every store in flight)
    q_k = FWD ? ws.tmp[1] : ws.tmp[0];
    dump = reinterpret_cast<u32*>(ws.sort_keys) + 2 * lane + (FWD ? 1 : 0);
    cap = list_cap;
    correct_lef_bar = lef_bar;
    do_secondary = secondary;
    // rev: followers i = max(1, n5) .. n-1 ascending, blocker = rank i-1
    // fwd: followers i-1 for i = (n - min(n3, n3-1) - 1) .. 1 descending, blocker = rank i
    f_first = FWD ? static_cast<i32>(bc.n3 == 0 ? n - 1 : n - bc.n3) - 1
                  : static_cast<i32>(umax(1u, bc.n5));
    nbatch = (n + 63) / 64;
    n_cand = 0;
    carry_pos = 0;
    carry_coll = 0;
    carry_pending = false;
    load_units(true, 0, cur);
  }
  // one group of UX batches (bg = first batch of the group)
  MODLE_DEV_MEMBER void step(u32 bg) {
    UnitRegs g = cur;
    load_units(false, bg, g);  // (defaults of the lanes outside the range)
    if (bg + UX < nbatch) load_units(true, bg + UX, cur);
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 bi = bg + u;
      if (bi >= nbatch) break;
      const i32 kk = FWD ? static_cast<i32>(n) - 1 - static_cast<i32>(bi * 64 + lane)
                         : static_cast<i32>(bi * 64 + lane);
      const bool act = kk >= 0 && static_cast<u32>(kk) < n;
      const u32 k = act ? static_cast<u32>(kk) : 0;
      const u32 P = g.P[u], M0 = g.M[u], C = g.C[u];
      u32 M = M0;
      if (correct_lef_bar && act && cw_occurred_as(C, EV_LEF_BAR)) {
        const u32 bp = g.B[u];
        M = (FWD ? bp - P : P - bp) - 1;
      }
      *((act && M != M0) ? &moves[k] : dump) = M;
      const bool follower = do_secondary && act && (FWD ? (kk <= f_first) : (kk >= f_first));
      // the blocker: the unit visited before this one
      const u32 bP_in = wave::shfl_up1(P), bC_in = wave::shfl_up1(C);
      const u32 bP = lane > 0 ? bP_in : carry_pos, bC = lane > 0 ? bC_in : carry_coll;
      const bool pot = follower && !cw_occurred(C) &&
                       (FWD ? static_cast<u64>(P) + M >= bP : static_cast<u64>(P) - M <= bP);
      const u64 potm = wave::ballot(pot);
      // blocker stalled already, or itself a candidate (then it may become stalled in pass 2):
      // propagate along runs of consecutive candidates.  The unit before lane 0 counts as "may be
      // stalled" when it is a candidate (its outcome is not known in this pass).
      const u64 occm = wave::ballot(cw_occurred(bC)) | (carry_pending ? u64(1) : u64(0));
      u64 pend = potm & occm;
      for (;;) {
        const u64 grown = pend | (potm & (pend << 1));
        if (grown == pend) break;
        pend = grown;
      }
      {
        const bool mine = ((pend >> lane) & 1u) != 0;
        const u32 e = n_cand + static_cast<u32>(wave::popc64(pend & lanemask_lt(lane)));
        const bool cont = lane > 0 ? ((pend >> (lane - 1)) & 1u) != 0 : carry_pending;
        *((mine && e < cap) ? &q_k[e] : dump) =
            k | (cont ? SEC_CONT : 0u) | (cw_occurred(bC) ? SEC_BOCC : 0u);
      }
      n_cand += static_cast<u32>(wave::popc64(pend));
      carry_pending = (pend >> 63) != 0;
      carry_pos = wave::bcast(P, 63);
      carry_coll = wave::bcast(C, 63);
    }
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
  const u32* ids = FWD ? ws.f_id : ws.r_id;
  u32* moves = FWD ? ws.f_move : ws.r_move;
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
#ifdef MODLE_PHASE_TIMERS
      const u64 t_walk = wave::clock();
#endif
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
#ifdef MODLE_PHASE_TIMERS
      c.ph[15] += wave::clock() - t_walk;
#endif
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
  for (u32 bg = 0; bg < f.nbatch; bg += SecondaryFilter<FWD>::UX) f.step(bg);
  wave::sync_mem();
  return secondary_resolve<FWD>(c, f.n_cand, list, list_cap, overflow);
}

// both directions: the two filters in one loop, then the rev and the fwd resolve pass (draw order)
MODLE_DEV_NOINLINE void process_secondary_both(Cell& c, BoundaryCounts bc, u32* list_rev, u32* list_fwd,
                                               u32 list_cap, bool& overflow, u32& n_rev, u32& n_fwd) {
  SecondaryFilter<false> fr;
  SecondaryFilter<true> ff;
  fr.init(c, bc, list_cap, true, true);
  ff.init(c, bc, list_cap, true, true);
#ifdef MODLE_PHASE_TIMERS
  const u64 t_pass1 = wave::clock();
#endif
  for (u32 bg = 0; bg < fr.nbatch; bg += SecondaryFilter<false>::UX) {
    fr.step(bg);
    ff.step(bg);
  }
  wave::sync_mem();
#ifdef MODLE_PHASE_TIMERS
  c.ph[14] += wave::clock() - t_pass1;  // (sub_a: the filter pass; the rest of the phase is pass 2)
#endif
  n_rev = secondary_resolve<false>(c, fr.n_cand, list_rev, list_cap, overflow);
  n_fwd = secondary_resolve<true>(c, ff.n_cand, list_fwd, list_cap, overflow);
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
    const u32 np1 = umin(ws.f_pos[ws.f_rank[id1]], p2);
    const u32 np2 = umin(ws.f_pos[ws.f_rank[id2]], p1);
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
    const u32 np1 = umax(ws.r_pos[ws.r_rank[id1]], p2);
    const u32 np2 = umax(ws.r_pos[ws.r_rank[id2]], p1);
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
        const u32 np1 = umin(ws.f_pos[ws.f_rank[id1]], p2);
        const u32 np2 = umin(ws.f_pos[ws.f_rank[id2]], p1);
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
        const u32 np1 = umax(ws.r_pos[ws.r_rank[id1]], p2);
        const u32 np2 = umax(ws.r_pos[ws.r_rank[id2]], p1);
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
  want_r = want_r && !c.inv_valid[0];
  want_f = want_f && !c.inv_valid[1];
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
  PHASE(c, 9, detect_lef_bar<false>(c, bc); detect_lef_bar<true>(c, bc));
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
  const u32 disp_cap = umin(STAGE_CAP, ws.capacity_lefs / 2);
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
          const wave::U32x4 zero = {{0, 0, 0, 0}};
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
        const wave::U32x4 zero = {{0, 0, 0, 0}};
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

// =============================================================================================
// Contact sampling (reference: src/libmodle/cpu/register_contacts.cpp)
// =============================================================================================
MODLE_DEV void matrix_increment(const Interval& iv, u64 row, u64 col) {
  // reference: contact_matrix_internal_impl.hpp:19-42, contact_matrix_dense_safe_impl.hpp:55-68
  u64 i, j;
  if (row > col) {
    i = row - col;
    j = row;
  } else {
    i = col - row;
    j = col;
  }
  if (i >= iv.nrows) {
    wave::atomic_add_u64(iv.missed_updates, 1);
  } else {
    wave::atomic_inc_u32(iv.contacts + (j * iv.nrows + i));
  }
}

enum EventKind { EV_LOOP = 0, EV_TAD = 1, EV_OCC = 2 };

struct EventEval {
  u32 consumed;     // raws consumed by the event when no draw was rejected
  bool need_exact;  // a rejection happened: the consumption is not known without a replay
  bool ok;          // the event yields a registration
  u64 a, b;         // the two genomic coordinates to register
};

// lef_within_bound (reference: register_contacts.cpp:23-29); returns the unit positions
MODLE_DEV bool lef_samplable(const Cell& c, u32 i, u32& rev, u32& fwd) {
  const Workspace& ws = c.ws;
  const u32 lo = c.iv->start + 1, hi = c.iv->end - 1;
  if (ws.epoch[i] == UNBOUND) return false;
  rev = ws.r_pos[ws.r_rank[i]];
  fwd = ws.f_pos[ws.f_rank[i]];
  return rev > lo && rev < hi && fwd > lo && fwd < hi;
}

// randomize_extrusion_unit_positions / pos_within_bound (reference: register_contacts.cpp:31-63)
MODLE_DEV bool sample_lef_pair(const Cell& c, u32 rev, u32 fwd, f64 u1, f64 u2, bool noisify,
                               f64& p1, f64& p2) {
  const Params& p = *c.p;
  const f64 n1 = noisify ? genextreme_from_canonical(u1, p.gev_mu, p.gev_sigma, p.gev_xi) : 0.0;
  const f64 a = static_cast<f64>(rev) - n1;
  const f64 n2 = noisify ? genextreme_from_canonical(u2, p.gev_mu, p.gev_sigma, p.gev_xi) : 0.0;
  const f64 b = static_cast<f64>(fwd) + n2;
  p1 = b < a ? b : a;
  p2 = b < a ? a : b;
  const f64 lo = static_cast<f64>(c.iv->start + 1), hi = static_cast<f64>(c.iv->end - 1);
  return p1 >= lo && p2 >= lo && p1 < hi && p2 < hi;
}

template <int KIND>
MODLE_DEV EventEval eval_event_fast(const Cell& c, u64 q, u64 lef_range, u64 lef_bucket,
                                    bool noisify) {
  EventEval e{1, false, false, 0, 0};
  const u64 r = rng_peek(c.g, q) / lef_bucket;
  if (r > lef_range) {
    e.need_exact = true;
    return e;
  }
  u32 rev = 0, fwd = 0;
  if (!lef_samplable(c, static_cast<u32>(r), rev, fwd)) return e;  // consumed = 1
  const u32 nz = noisify ? 2u : 0u;
  const f64 u1 = noisify ? canonical_raw(rng_peek(c.g, q + 1)) : 0.0;
  const f64 u2 = noisify ? canonical_raw(rng_peek(c.g, q + 2)) : 0.0;
  f64 p1, p2;
  const bool inb = sample_lef_pair(c, rev, fwd, u1, u2, noisify, p1, p2);
  e.consumed = 1 + nz;
  if (!inb) return e;
  const u64 a = static_cast<u64>(p1), b = static_cast<u64>(p2);
  if (KIND != EV_TAD) {
    e.ok = true;
    e.a = a;
    e.b = b;
    return e;
  }
  const u64 range = b - a;
  if (range == 0) {
    e.ok = true;
    e.a = a;
    e.b = a;
    return e;
  }
  const u64 bucket = uniform_int_bucket(range);
  const u64 ra = rng_peek(c.g, q + 1 + nz) / bucket;
  const u64 rb = rng_peek(c.g, q + 2 + nz) / bucket;
  if (ra > range || rb > range) {
    e.need_exact = true;
    return e;
  }
  e.consumed = 3 + nz;
  e.ok = true;
  e.a = a + ra;
  e.b = a + rb;
  return e;
}

// one sampling event replayed sequentially from g.pos; uniform
template <int KIND>
MODLE_DEV_NOINLINE EventEval eval_event_exact(Cell& c, u64 lef_range, u64 lef_bucket, bool noisify) {
  EventEval e{0, false, false, 0, 0};
  const u64 r = lef_range == 0 ? 0 : uniform_int_exact(c.g, lef_range, lef_bucket);
  u32 rev = 0, fwd = 0;
  if (!lef_samplable(c, static_cast<u32>(r), rev, fwd)) return e;
  const f64 u1 = noisify ? canonical_raw(rng_next(c.g)) : 0.0;
  const f64 u2 = noisify ? canonical_raw(rng_next(c.g)) : 0.0;
  f64 p1, p2;
  if (!sample_lef_pair(c, rev, fwd, u1, u2, noisify, p1, p2)) return e;
  const u64 a = static_cast<u64>(p1), b = static_cast<u64>(p2);
  e.ok = true;
  if (KIND != EV_TAD) {
    e.a = a;
    e.b = b;
    return e;
  }
  const u64 range = b - a;
  if (range == 0) {
    e.a = a;
    e.b = a;
    return e;
  }
  const u64 bucket = uniform_int_bucket(range);
  e.a = a + uniform_int_exact(c.g, range, bucket);
  e.b = a + uniform_int_exact(c.g, range, bucket);
  return e;
}

template <int KIND>
MODLE_DEV void commit_event(const Cell& c, const EventEval& e) {
  const Interval& iv = *c.iv;
  const u64 lo = static_cast<u64>(iv.start) + 1;
  const u64 bin = c.p->bin_size;
  const u64 ba = (e.a - lo) / bin, bb = (e.b - lo) / bin;
  if (KIND == EV_OCC) {
    if (iv.occupancy_1d != nullptr) {
      wave::atomic_add_u64(iv.occupancy_1d + ba, 1);
      wave::atomic_add_u64(iv.occupancy_1d + bb, 1);
    }
  } else {
    matrix_increment(iv, ba, bb);
  }
}

// runs `n_events` sampling events of one kind; returns the number of registrations
template <int KIND>
MODLE_DEV_NOINLINE u64 run_events(Cell& c, u64 n_events) {
  if (n_events == 0) return 0;
  const u32 lane = wave::lane();
  const bool noisify = (c.p->sampling_strategy & CS_NOISIFY) != 0;
  const u64 lef_range = static_cast<u64>(c.n_active) - 1;
  const u64 lef_bucket = lef_range != 0 ? uniform_int_bucket(lef_range) : 1;
  const u32 stride = 1 + (noisify ? 2u : 0u) + (KIND == EV_TAD ? 2u : 0u);
  u64 registered = 0;
  u64 remaining = n_events;
  while (remaining != 0) {
    if (lef_range == 0) {
      // a single LEF: the index draw consumes nothing; keep it simple and replay sequentially
      const EventEval e = eval_event_exact<KIND>(c, lef_range, lef_bucket, noisify);
      if (e.ok && lane == 0) commit_event<KIND>(c, e);
      registered += e.ok ? 1 : 0;
      --remaining;
      continue;
    }
    // one step handles at most as many events as the PRNG ring can serve
    const u32 cntb = static_cast<u32>(umin64(umin(64u, RNG_BLOCK / stride), remaining));
    rng_ensure(c.g, cntb * stride);
    const bool act = lane < cntb;
    EventEval e{stride, false, false, 0, 0};
    if (act) e = eval_event_fast<KIND>(c, c.g.pos + static_cast<u64>(lane) * stride, lef_range,
                                       lef_bucket, noisify);
    const u64 irregular = wave::ballot(act && (e.need_exact || e.consumed != stride));
    if (irregular == 0) {
      if (act && e.ok) commit_event<KIND>(c, e);
      registered += static_cast<u64>(wave::popc64(wave::ballot(act && e.ok)));
      rng_advance(c.g, static_cast<u64>(cntb) * stride);
      remaining -= cntb;
    } else {
      const u32 f = static_cast<u32>(wave::ctz64(irregular));
      const bool commit = act && e.ok && lane < f;
      if (commit) commit_event<KIND>(c, e);
      registered += static_cast<u64>(wave::popc64(wave::ballot(commit)));
      rng_advance(c.g, static_cast<u64>(f) * stride);
      const bool needx = wave::bcast(e.need_exact, f);
      if (!needx) {
        if (lane == f && e.ok) commit_event<KIND>(c, e);
        registered += wave::bcast(e.ok, f) ? 1 : 0;
        rng_advance(c.g, wave::bcast(e.consumed, f));
      } else {
        const EventEval x = eval_event_exact<KIND>(c, lef_range, lef_bucket, noisify);
        if (x.ok && lane == 0) commit_event<KIND>(c, x);
        registered += x.ok ? 1 : 0;
      }
      remaining -= f + 1;
    }
  }
  return registered;
}

// sample_and_register_contacts (reference: register_contacts.cpp:93-120)
MODLE_DEV u64 phase_sample_contacts(Cell& c, u64 events_per_epoch, u64 num_target_contacts,
                                    u64 num_contacts, u64& events_done) {
  const Params& p = *c.p;
  u64 n_events = events_per_epoch;
  if (p.target_contact_density > 0.0)
    n_events = umin64(n_events, num_target_contacts - num_contacts);
  if (n_events == 0) return 0;
  ensure_inverse_both(c);  // events pick LEFs by id
  events_done += n_events;
  u64 n_loop;
  if (p.tad_to_loop_ratio == 0) {
    n_loop = n_events;
  } else if (!wave::f_isfinite(p.tad_to_loop_ratio)) {
    n_loop = 0;
  } else {
    n_loop = static_cast<u64>(
        binomial_exact(c.g, static_cast<i64>(n_events), 1.0 / (p.tad_to_loop_ratio + 1.0)));
  }
  u64 registered = run_events<EV_LOOP>(c, n_loop);
  registered += run_events<EV_TAD>(c, n_events - n_loop);
  if (p.track_1d) (void)run_events<EV_OCC>(c, n_events);
  return registered;
}

// =============================================================================================
// Burn-in (reference: simulation.cpp:795-894)
// =============================================================================================
struct LoopStats {
  f64 avg, std;  // stats::mean / stats::standard_dev of the loop sizes (population std)
};
MODLE_DEV LoopStats loop_size_stats(Cell& c) {
  // reference: simulation.cpp:795-819 and stats/descriptive_impl.hpp:22-31, 63-101.  The mean is
  // a sum of integers below 2^53 (order independent); the squared deviations are accumulated
  // strictly left to right in LEF-id order like std::accumulate.
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  // pass A, rank order (contiguous reads, four batches in flight): every unit drops its position
  // at its LEF's slot of two id-ordered scratch arrays; the sum of all loop sizes is the sum of
  // the fwd positions minus the sum of the rev positions (released LEFs have both units at
  // UNBOUND and cancel: loop size 0, like the reference)
  u32* by_id_fwd = ws.tmp[0];
  u32* by_id_rev = ws.tmp[1];
  constexpr u32 UX = 4;  // batches per group; the next group's loads go before this group's stores
  u64 part = 0;
  struct UnitRegs {
    u32 fP[UX], fI[UX], rP[UX], rI[UX];
  };
  const auto load_units = [&](auto op, u32 group, UnitRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 k = group + 64 * u + lane;
      const bool act = k < n;
      r.fP[u] = op(ws.f_pos, k, act, 0, r.fP[u]);
      r.fI[u] = op(ws.f_id, k, act, 0, r.fI[u]);
      r.rP[u] = op(ws.r_pos, k, act, 0, r.rP[u]);
      r.rI[u] = op(ws.r_id, k, act, 0, r.rI[u]);
    }
  };
  UnitRegs cur;
  load_units(wave::LdRaw{}, 0, cur);
  for (u32 group = 0; group < n; group += 64 * UX) {
    UnitRegs g = cur;
    load_units(wave::LdMask{}, group, g);  // (defaults of the lanes outside the range)
    if (group + 64 * UX < n) load_units(wave::LdRaw{}, group + 64 * UX, cur);
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 k = group + 64 * u + lane;
      if (k < n) {
        by_id_fwd[g.fI[u]] = g.fP[u];
        by_id_rev[g.rI[u]] = g.rP[u];
        part += static_cast<u64>(g.fP[u]) - static_cast<u64>(g.rP[u]);
      }
    }
  }
  wave::sync_mem();
#pragma unroll
  for (u32 s = 1; s < 64; s <<= 1) {
    const u64 o = wave::shfl_down(part, s);
    if (lane + s < 64) part += o;
  }
  const u64 total = wave::bcast(part, 0);
  const f64 avg = static_cast<f64>(total) / static_cast<f64>(n);
  // pass B, LEF-id order: strictly sequential accumulation like std::accumulate: every lane
  // computes its term, the terms of a batch are folded in lane order through broadcasts
  f64 ssd = 0.0;
  f64* terms = reinterpret_cast<f64*>(c.lds.stage);  // 2 x 64 terms (the buffer is idle here)
  static_assert(STAGE_CAP * sizeof(u32) >= 128 * sizeof(f64), "stage buffer too small for the fold");
  struct SizeRegs {
    u32 lf[UX], lr[UX];
  };
  const auto load_sizes = [&](u32 group, SizeRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 i = group + 64 * u + lane;
      r.lf[u] = wave::LdRaw{}(by_id_fwd, i, i < n, 0, 0u);
      r.lr[u] = wave::LdRaw{}(by_id_rev, i, i < n, 0, 0u);
    }
  };
  SizeRegs scur;
  load_sizes(0, scur);
  for (u32 group = 0; group < n; group += 64 * UX) {
    const SizeRegs sg = scur;  // (the next group's loads are in flight during the fold)
    if (group + 64 * UX < n) load_sizes(group + 64 * UX, scur);
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 base = group + 64 * u;
      if (base >= n) break;
      const u32 i = base + lane;
      f64 term = 0.0;
      if (i < n) {
        const u32 ls = sg.lf[u] - sg.lr[u];
        const f64 d = static_cast<f64>(static_cast<u64>(ls)) - avg;
        term = d * d;
      }
      // lanes past the end hold +0.0, which leaves the (non-negative) running sum unchanged, so
      // all 64 terms are folded with constant indices (no loop control in the chain).  The terms
      // go through LDS: every lane reads them back in order (one address for the whole wave: a
      // broadcast) and keeps its own copy of the running sum.  Two lane broadcasts per term plus
      // the wait states between a broadcast and the addition that uses it had been two thirds of
      // the chain.
      wave::lockstep();
      terms[64 * (u & 1u) + lane] = term;
      wave::sync_lds();
#pragma unroll
      for (u32 l = 0; l < 64; ++l) ssd = ssd + terms[64 * (u & 1u) + l];
    }
  }
  return LoopStats{avg, wave::f_sqrt(ssd / static_cast<f64>(n))};
}

MODLE_DEV_NOINLINE void compute_loop_size_stats(Cell& c) {
  Workspace& ws = c.ws;
  const u32 lane = wave::lane();
  const u32 cap = c.p->hist_len;
  const LoopStats st = loop_size_stats(c);
  const f64 avg = st.avg, std = st.std;
  // push_back with pop_front at capacity (two deque<double>)
  f64* cfx = ws.hist;
  f64* avgb = ws.hist + cap;
  u32 slot;
  if (c.hist_len == cap) {
    slot = c.hist_head;
    c.hist_head = (c.hist_head + 1) % cap;
  } else {
    slot = (c.hist_head + c.hist_len) % cap;
    ++c.hist_len;
  }
  wave::lockstep();
  if (lane == 0) {
    avgb[slot] = avg;
    cfx[slot] = std / avg;
  }
  wave::sync_mem();
}

MODLE_DEV bool series_is_stable(const Cell& c, const f64* buf) {
  const u32 cap = c.p->hist_len, w = c.p->window;
  const u32 lane = wave::lane();
  const u32 ncmp = cap - w - 1;  // comparisons of consecutive window means
  u32 n_dips = 0;
  for (u32 base = 0; base < ncmp; base += 64) {
    const u32 j = base + lane;
    bool dip = false;
    if (j < ncmp) {
      f64 s1 = 0.0, s2 = 0.0;
      for (u32 t = 0; t < w; ++t) s1 = s1 + buf[(c.hist_head + j + t) % cap];
      for (u32 t = 0; t < w; ++t) s2 = s2 + buf[(c.hist_head + j + 1 + t) % cap];
      dip = (s1 / static_cast<f64>(w)) > (s2 / static_cast<f64>(w));
    }
    n_dips += static_cast<u32>(wave::popc64(wave::ballot(dip)));
  }
  const f64 r = static_cast<f64>(n_dips) / static_cast<f64>(cap - w - n_dips);
  return r >= 0.95 && r <= 1.05;
}

MODLE_DEV_NOINLINE bool evaluate_burnin(const Cell& c) {
  // reference: simulation.cpp:821-864
  const u32 cap = c.p->hist_len;
  if (c.hist_len != cap) return false;
  if (!series_is_stable(c, c.ws.hist)) return false;
  return series_is_stable(c, c.ws.hist + cap);
}

// =============================================================================================
// Cell driver (reference: simulation.cpp:896-986)
// =============================================================================================
MODLE_DEV_NOINLINE void reset_cell_buffers(Cell& c) {
  // State::reset_buffers (reference: simulation.cpp:617-627)
  Workspace& ws = c.ws;
  const u32 L = wave::uniform(c.n_lefs);
  const u32 lane = wave::lane();
  for (u32 base = 0; base < L; base += 64) {
    const u32 i = base + lane;
    if (i < L) {
      ws.r_pos[i] = UNBOUND;
      ws.f_pos[i] = UNBOUND;
      ws.epoch[i] = UNBOUND;
      ws.r_id[i] = i;
      ws.f_id[i] = i;
      ws.r_rank[i] = i;
      ws.f_rank[i] = i;
      ws.r_move[i] = 0;
      ws.f_move[i] = 0;
      ws.r_coll[i] = 0;
      ws.f_coll[i] = 0;
      ws.stall[i] = 0;
    }
  }
  wave::sync_mem();
}

// LEFs n_old .. n_new-1 become active.  They have never been ranked: their slots are the
// identity (the reference's iota-initialised rank buffers, simulation.cpp:617-620); the id
// arrays are re-initialised here because they are double-buffered by rank_update.
MODLE_DEV void activate_lefs(Cell& c, u32 n_old, u32 n_new) {
  const u32 lane = wave::lane();
  for (u32 base = n_old; base < n_new; base += 64) {
    const u32 k = base + lane;
    if (k < n_new) {
      c.ws.r_id[k] = k;
      c.ws.f_id[k] = k;
      c.ws.r_rank[k] = k;
      c.ws.f_rank[k] = k;
    }
  }
  c.n_active = n_new;
  wave::sync_mem();
}

// copy of an interval descriptor whose pointers are known to address device memory
MODLE_DEV Interval interval_in_device_memory(const Interval& iv) {
  Interval g = iv;
  g.bar_pos = wave::as_global(iv.bar_pos);
  g.bar_dir = wave::as_global(iv.bar_dir);
  g.bar_stp_active = wave::as_global(iv.bar_stp_active);
  g.bar_stp_inactive = wave::as_global(iv.bar_stp_inactive);
  g.bar_occupancy = wave::as_global(iv.bar_occupancy);
  g.contacts = wave::as_global(iv.contacts);
  g.occupancy_1d = wave::as_global(iv.occupancy_1d);
  g.missed_updates = wave::as_global(iv.missed_updates);
  g.bar_bucket = wave::as_global(iv.bar_bucket);
  return g;
}

MODLE_DEV void init_cell(Cell& c, const Params& p, const Interval& iv, const Workspace& ws,
                         const WaveLds& lds, u32 n_lefs, const u64 prng[4]) {
  c.p = &p;
  c.iv = &iv;
  c.ws = ws;
  c.lds = lds;
  c.n_lefs = n_lefs;
  c.n_active = 0;
  c.hist_len = 0;
  c.hist_head = 0;
  c.error = 0;
  c.n_hit[0] = 0;
  c.n_hit[1] = 0;
  c.n_rel = 0;
  c.rel_valid = false;  // the epoch loop turns the list on; the phase-level hooks sweep
  c.keys_valid = false;
  c.n_keys = 0;
  c.n_disp[0] = 0;
  c.n_disp[1] = 0;
  c.disp_valid = true;  // (nothing has been ranked yet: nothing can be out of order)
  c.n_bound = 0;
  c.inv_valid[0] = true;  // (reset_cell_buffers / run_test_phases write complete permutations)
  c.inv_valid[1] = true;
  c.filter_on = false;
#ifdef MODLE_PHASE_TIMERS
  for (int i = 0; i < 16; ++i) c.ph[i] = 0;
#endif
  c.g.ring = lds.ring;
  c.g.jump = lds.jump_table;
  c.g.state = lds.rng_state;
  c.g.snap = lds.rng_snap;
  rng_init(c.g, prng);
}

// Diagnostic trace (enabled by the host with MODLE_HIP_TRACE_SHM): after selected phases of every
// epoch, order-sensitive checksums of the unit arrays and the PRNG position are stored, so that
// a run on the GPU can be compared phase by phase with a run under the CPU lane emulator.
constexpr u32 TRACE_STAGES = 8;
constexpr u32 TRACE_WORDS_PER_STAGE = 6;
// Compiled in only with MODLE_STAGE_TRACE (`make trace`, the emulator build): seven inlined copies
// of this function are a seventh of the kernel's code, all of it dead weight in the instruction
// cache of a normal run.
#ifndef MODLE_STAGE_TRACE
MODLE_DEV void trace_stage(Cell&, u64, u32) {}
#else
MODLE_DEV_NOINLINE void trace_stage(Cell& c, u64 epoch, u32 stage) {
  u64* tr = c.lds.trace;
  if (tr == nullptr || epoch >= c.lds.trace_cap) return;
  const u32 lane = wave::lane();
  u64 s[4] = {0, 0, 0, 0};
  for (u32 base = 0; base < c.n_active; base += 64) {
    const u32 k = base + lane;
    if (k < c.n_active) {
      const u64 w = k + 1;
      s[0] += w * c.ws.r_pos[k] + c.ws.r_id[k];
      s[1] += w * c.ws.f_pos[k] + c.ws.f_id[k];
      s[2] += w * c.ws.r_move[k];
      s[3] += w * c.ws.f_move[k];
    }
  }
#pragma unroll
  for (u32 q = 0; q < 4; ++q) {
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
      const u64 o = wave::shfl_down(s[q], d);
      if (lane + d < 64) s[q] += o;
    }
  }
  if (lane == 0) {
    u64* rec = tr + (epoch * TRACE_STAGES + stage) * TRACE_WORDS_PER_STAGE;
    rec[0] = c.g.pos;
    rec[1] = s[0];
    rec[2] = s[1];
    rec[3] = s[2];
    rec[4] = s[3];
    rec[5] = (static_cast<u64>(c.n_active) << 32) | (stage + 1);
  }
}
#endif

// Model-internal-state record of one epoch (Simulation::dump_stats, reference:
// simulation.cpp:995-1056; logged after extrude and before release_lefs, :969-975).  Compiled in
// only with MODLE_STATE_LOG (`make statelog`): the front end loads that build when
// --log-model-internal-state is given.  Runs BEFORE the fused extrusion / release pass (which
// consumes the collision words), on positions + moves = the positions after extrusion.
#ifdef MODLE_STATE_LOG
MODLE_DEV_NOINLINE void log_internal_state(Cell& c, u64 epoch, bool burnin) {
  u64* log = c.lds.state_log;
  if (log == nullptr || epoch >= c.lds.state_log_cap) return;
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  u32* flag = ws.tmp[2];  // per LEF: its rev unit is stalled
  u32 st_rev = 0, st_fwd = 0, st_both = 0, n_bar = 0, n_prim = 0, n_sec = 0;
  u64 part = 0;
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    const bool act = k < n;
    const u32 rc = wave::ld_sel(ws.r_coll, k, act, 0);
    const u32 P = wave::ld_sel(ws.r_pos, k, act, UNBOUND);
    if (act) flag[ws.r_id[k]] = cw_occurred(rc) ? 1u : 0u;
    st_rev += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred(rc))));
    n_bar += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred_as(rc, EV_LEF_BAR))));
    n_prim += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred_as(rc, EV_LEF_LEF_PRIMARY))));
    n_sec += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred_as(rc, EV_LEF_LEF_SECONDARY))));
    if (act && P != UNBOUND) part -= static_cast<u64>(P - ws.r_move[k]);
  }
  wave::sync_mem();
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    const bool act = k < n;
    const u32 fc = wave::ld_sel(ws.f_coll, k, act, 0);
    const u32 P = wave::ld_sel(ws.f_pos, k, act, UNBOUND);
    const bool both = act && cw_occurred(fc) && flag[ws.f_id[k]] != 0;
    st_fwd += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred(fc))));
    st_both += static_cast<u32>(wave::popc64(wave::ballot(both)));
    n_bar += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred_as(fc, EV_LEF_BAR))));
    n_prim += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred_as(fc, EV_LEF_LEF_PRIMARY))));
    n_sec += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred_as(fc, EV_LEF_LEF_SECONDARY))));
    if (act && P != UNBOUND) part += static_cast<u64>(P + ws.f_move[k]);
  }
#pragma unroll
  for (u32 sft = 1; sft < 64; sft <<= 1) {
    const u64 o = wave::shfl_down(part, sft);
    if (lane + sft < 64) part += o;
  }
  const u64 loop_sum = wave::bcast(part, 0);
  u32 n_occ = 0;
  const u32 nb = wave::uniform(c.iv->n_barriers);
  for (u32 base = 0; base < nb; base += 64) {
    const u32 i = base + lane;
    n_occ += static_cast<u32>(wave::popc64(wave::ballot(i < nb && ws.bar_active[i] != 0)));
  }
  if (lane == 0) {
    u64* rec = log + epoch * STATE_LOG_WORDS;
    rec[0] = epoch | (burnin ? (u64(1) << 63) : 0);
    rec[1] = n_occ;
    rec[2] = n;
    rec[3] = st_rev;
    rec[4] = st_fwd;
    rec[5] = st_both;
    rec[6] = n_bar;
    rec[7] = n_prim;
    rec[8] = n_sec;
    rec[9] = loop_sum;
  }
  wave::sync_mem();
}
#else
MODLE_DEV void log_internal_state(Cell&, u64, bool) {}
#endif

// Simulates one (interval, cell) task on the calling wave.  Returns 0 or a non-zero status when
// an internal capacity was exceeded (the host turns that into an error).
MODLE_DEV u32 simulate_cell(const Params& p, const Interval& iv, const Task& task,
                            const Workspace& ws, const WaveLds& lds, CellResult& res) {
  Cell c;
  const Interval ivg = interval_in_device_memory(iv);
  init_cell(c, p, ivg, ws, lds, task.num_lefs, task.prng);
  reset_cell_buffers(c);

  u64 epoch = 0, num_burnin_epochs = 0, num_contacts = 0;
  u64 sum_active = 0, events_done = 0, sim_epochs = 0;
  bool burnin_completed = false;
  u32 status = 0;
  const f64 lef_binding_rate_burnin =
      static_cast<f64>(task.num_lefs) / static_cast<f64>(p.burnin_target_epochs_for_lef_activation);

  barriers_init_states(c);
  c.rel_valid = true;  // nothing released yet: every LEF to bind is a newly activated one
  if (p.skip_burnin) {
    activate_lefs(c, 0, c.n_lefs);
    burnin_completed = true;
  }
  for (;; ++epoch) {
    if (p.target_contact_density >= 0) {
      if (num_contacts >= task.num_target_contacts) break;
    } else if (epoch - num_burnin_epochs >= task.num_target_epochs) {
      break;
    }
    // cancellation, checked once per epoch like the reference's `_ctx` (simulation.cpp:933)
    if (lds.abort_flag != nullptr && wave::uniform(wave::load_agent_u32(lds.abort_flag)) != 0) {
      status = ERR_CANCELLED;
      break;
    }
    if (!burnin_completed) {
      // run_burnin (reference: simulation.cpp:866-894)
      do {
        ++num_burnin_epochs;
        if (c.n_active != c.n_lefs) {
          const u64 k = poisson_exact(c.g, lef_binding_rate_burnin);
          const u64 na = static_cast<u64>(c.n_active) + k;
          activate_lefs(c, c.n_active, na < c.n_lefs ? static_cast<u32>(na) : c.n_lefs);
        } else {
          PHASE(c, 0, compute_loop_size_stats(c); burnin_completed = evaluate_burnin(c));
          burnin_completed = burnin_completed && epoch > p.min_burnin_epochs;
          if (!burnin_completed && epoch >= p.max_burnin_epochs) {
            burnin_completed = true;
            activate_lefs(c, c.n_active, c.n_lefs);
          }
        }
      } while (c.n_active == 0);
    }
    PHASE(c, 1, if (c.rel_valid) phase_bind_listed(c, static_cast<u32>(epoch));
          else {
            phase_bind(c, static_cast<u32>(epoch));
            c.n_bound = c.n_active;
          });
    trace_stage(c, epoch, 0);
    PHASE(c, 2, rank_update<false>(c, false));
    PHASE(c, 3, rank_update<true>(c, false));
    trace_stage(c, epoch, 1);
    if (c.error != 0) {
      status = c.error;
      break;
    }

    if (burnin_completed) {
      PHASE(c, 4, num_contacts += phase_sample_contacts(c, task.contacts_per_epoch,
                                                        task.num_target_contacts, num_contacts,
                                                        events_done));
      trace_stage(c, epoch, 5);
      if (task.num_target_contacts != 0 && num_contacts >= task.num_target_contacts) break;
    }

    sum_active += c.n_active;
    ++sim_epochs;
    phase_generate_moves(c, burnin_completed);
    trace_stage(c, epoch, 2);
    PHASE(c, 7, barriers_next_state(c));
    const bool coll_ok = phase_process_collisions(c);
    trace_stage(c, epoch, 3);
    if (!coll_ok) {
      status = c.error;
      break;
    }
    log_internal_state(c, epoch, !burnin_completed);
    PHASE(c, 13, phase_extrude_and_release(c, burnin_completed));
    trace_stage(c, epoch, 4);
  }

  trace_stage(c, epoch, 6);
#ifdef MODLE_PHASE_TIMERS
  if (lds.phase_ticks != nullptr && wave::lane() == 0) {
    for (int i = 0; i < 16; ++i) wave::atomic_add_u64(lds.phase_ticks + i, c.ph[i]);
  }
  wave::lockstep();
#endif
  res.epochs = epoch;
  res.burnin_epochs = num_burnin_epochs;
  res.num_contacts = num_contacts;
  res.raws_consumed = c.g.pos;
  rng_final_state(c.g, res.prng_final);
  res.sum_active_lefs = sum_active;
  res.sampling_events = events_done;
  res.sim_epochs = sim_epochs;
  return status;
}

// =============================================================================================
// Phase-level entry point (mirrors Simulation::test_* hooks, reference: simulation.hpp:413-567)
//
// The caller's arrays use the reference's layout: positions / binding epochs / moves / collision
// words indexed by LEF id plus two rank arrays (LEF id at every rank).  They arrive in scratch
// arrays (`TestImage`), are converted to the rank-ordered device layout, the requested passes
// run, and the result is converted back.
// =============================================================================================
constexpr u32 PH_RANK = 0x001, PH_RANK_INIT = 0x002, PH_ADJUST = 0x004, PH_CLAMP = 0x008,
              PH_BOUNDARIES = 0x010, PH_LEF_BAR = 0x020, PH_PRIMARY = 0x040,
              PH_CORRECT_LEF_BAR = 0x080, PH_CORRECT_PRIMARY = 0x100, PH_SECONDARY = 0x200,
              PH_FIX_SECONDARY = 0x400, PH_USE_BOUNDARY_COUNTS = 0x800,
              PH_BIND = 0x1000,       // select_and_bind_lefs: bind every released LEF, then rank
              PH_GEN_MOVES = 0x2000;  // generate_moves: draw, adjust, clamp
// bits 16..31 of the mask: the current epoch (binding epoch of the LEFs PH_BIND binds)

struct TestImage {  // all by LEF id except the two rank arrays; n entries each
  u32 *rev_pos, *fwd_pos, *epoch, *rev_rank, *fwd_rank, *rev_moves, *fwd_moves, *rev_coll,
      *fwd_coll;
};

// rank positions whose unit carries an "avoided secondary collision" mark, in the order
// process_secondary would have produced them
template <bool FWD>
MODLE_DEV u32 collect_avoided(Cell& c, u32* list, u32 cap) {
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  const u32* coll = FWD ? c.ws.f_coll : c.ws.r_coll;
  u32 cnt = 0;
  for (u32 base = 0; base < n; base += 64) {
    const u32 off = base + lane;
    const bool act = off < n;
    const u32 k = FWD ? (n - 1 - off) : off;
    const bool hit = act && cw_avoided_as(coll[act ? k : 0], EV_LEF_LEF_SECONDARY) &&
                     (FWD ? k + 1 < n : k >= 1);
    const u64 m = wave::ballot(hit);
    if (hit) {
      const u32 j = cnt + static_cast<u32>(wave::popc64(m & lanemask_lt(lane)));
      if (j < cap) list[j] = k;
    }
    cnt += static_cast<u32>(wave::popc64(m));
  }
  wave::sync_mem();
  return cnt < cap ? cnt : cap;
}

MODLE_DEV u32 run_test_phases(const Params& p, const Interval& iv, const Workspace& ws,
                              const WaveLds& lds, const TestImage& img, u32 mask, u32 n,
                              const u64 prng[4], u64& raws_consumed) {
  Cell c;
  const Interval ivg = interval_in_device_memory(iv);
  init_cell(c, p, ivg, ws, lds, n, prng);
  c.n_active = n;
  const u32 lane = wave::lane();
  // reference layout -> device layout
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    if (k < n) {
      const bool init = (mask & PH_RANK) && (mask & PH_RANK_INIT);
      const u32 rid = init ? k : img.rev_rank[k];
      const u32 fid = init ? k : img.fwd_rank[k];
      c.ws.r_id[k] = rid;
      c.ws.r_pos[k] = img.rev_pos[rid];
      c.ws.r_move[k] = img.rev_moves[rid];
      c.ws.r_coll[k] = img.rev_coll[rid];
      c.ws.r_rank[rid] = k;
      c.ws.f_id[k] = fid;
      c.ws.f_pos[k] = img.fwd_pos[fid];
      c.ws.f_move[k] = img.fwd_moves[fid];
      c.ws.f_coll[k] = img.fwd_coll[fid];
      c.ws.f_rank[fid] = k;
      c.ws.epoch[k] = img.epoch[k];
      c.ws.stall[k] = 0;
    }
  }
  wave::sync_mem();
  // barrier positions of LEF-BAR words that came with the image (detect_lef_bar writes them
  // itself when it runs)
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    if (k < n) {
      const u32 rc = c.ws.r_coll[k], fc = c.ws.f_coll[k];
      if (cw_occurred_as(rc, EV_LEF_BAR)) stalling_barrier_positions<false>(c.ws)[k] = stalling_barrier_pos(ivg, rc);
      if (cw_occurred_as(fc, EV_LEF_BAR)) stalling_barrier_positions<true>(c.ws)[k] = stalling_barrier_pos(ivg, fc);
    }
  }
  wave::sync_mem();
  if (mask & PH_BIND) {
    // Simulation::select_and_bind_lefs (simulation.cpp:988-993): the released LEFs of the image
    // are the ones to bind; the ranking that follows is the partially sorted one
    phase_bind(c, mask >> 16);
    rank_update<false>(c, false);
    rank_update<true>(c, false);
  }
  if (mask & PH_RANK) {
    // positions only: move / collision arrays are not meaningful across a re-ranking
    rank_update<false>(c, true);
    rank_update<true>(c, true);
  }
  if (mask & PH_GEN_MOVES) {
    bool unbound = false;
    for (u32 base = 0; base < n; base += 64) {
      const u32 k = base + lane;
      unbound = wave::any(k < n && c.ws.epoch[k] == UNBOUND) || unbound;
    }
    phase_generate_moves(c, true, !unbound);
  }
  if (mask & (PH_ADJUST | PH_CLAMP)) {
    adjust_moves_rev(c, (mask & PH_ADJUST) != 0, (mask & PH_CLAMP) != 0);
    adjust_moves_fwd(c, (mask & PH_ADJUST) != 0, (mask & PH_CLAMP) != 0);
  }
  BoundaryCounts bc{0, 0};
  if (mask & PH_BOUNDARIES) {
    const BoundaryCounts got = detect_boundaries(c);
    if (mask & PH_USE_BOUNDARY_COUNTS) bc = got;
  }
  if (mask & PH_LEF_BAR) {
    if (stalling_lists_wanted(p)) compact_stalling_barriers(c);
    detect_lef_bar<false>(c, bc);
    detect_lef_bar<true>(c, bc);
  }
  // the reference's hook sequences run "correct LEF-BAR moves" before "correct primary moves";
  // with both requested the fused forms are equivalent, otherwise run them stand-alone
  const bool fuse = (mask & PH_CORRECT_PRIMARY) && (mask & PH_CORRECT_LEF_BAR) && (mask & PH_PRIMARY);
  if (mask & PH_PRIMARY) detect_primary(c, bc, fuse);
  u32* list_rev = c.ws.tmp[5];
  u32* list_fwd = c.ws.tmp[6];
  const u32 cap = c.ws.capacity_lefs;
  u32 nr = 0, nf = 0;
  bool overflow = false;
  if (!fuse && (mask & PH_CORRECT_LEF_BAR)) {
    (void)process_secondary<false>(c, bc, list_rev, cap, overflow, true, false);
    (void)process_secondary<true>(c, bc, list_fwd, cap, overflow, true, false);
  }
  if (!fuse && (mask & PH_CORRECT_PRIMARY)) correct_moves_primary_standalone(c);
  if (mask & PH_SECONDARY) {
    nr = process_secondary<false>(c, bc, list_rev, cap, overflow, fuse, true);
    nf = process_secondary<true>(c, bc, list_fwd, cap, overflow, fuse, true);
  } else if (fuse) {
    (void)process_secondary<false>(c, bc, list_rev, cap, overflow, true, false);
    (void)process_secondary<true>(c, bc, list_fwd, cap, overflow, true, false);
  }
  if (mask & PH_FIX_SECONDARY) {
    if (!(mask & PH_SECONDARY)) {
      nr = collect_avoided<false>(c, list_rev, cap);
      nf = collect_avoided<true>(c, list_fwd, cap);
    }
    if (nr != 0) fix_secondary_rev(c, list_rev, nr);
    if (nf != 0) fix_secondary_fwd(c, list_fwd, nf);
  }
  // device layout -> reference layout
  wave::sync_mem();
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    if (k < n) {
      const u32 rid = c.ws.r_id[k], fid = c.ws.f_id[k];
      img.rev_rank[k] = rid;
      img.fwd_rank[k] = fid;
      img.rev_pos[rid] = c.ws.r_pos[k];
      img.rev_moves[rid] = c.ws.r_move[k];
      img.rev_coll[rid] = c.ws.r_coll[k];
      img.fwd_pos[fid] = c.ws.f_pos[k];
      img.fwd_moves[fid] = c.ws.f_move[k];
      img.fwd_coll[fid] = c.ws.f_coll[k];
      img.epoch[k] = c.ws.epoch[k];  // binding epochs change under PH_BIND
    }
  }
  wave::sync_mem();
  raws_consumed = c.g.pos;
  return overflow ? ERR_LIST_OVERFLOW : c.error;
}

// =============================================================================================
// Unit-level entry point: the small pieces of the path the reference tests on their own
// (test/units/stats/descriptive_test.cpp, test/units/contact_matrix/*_test.cpp,
// test/units/simulation_cpu/collision_encoding_test.cpp), run by the device code itself.
// =============================================================================================
constexpr u32 UNIT_LOOP_STATS = 1, UNIT_MATRIX_INCREMENT = 2, UNIT_COLLISION_WORDS = 3,
              UNIT_MATH_LOG_EXP = 4, UNIT_MATH_POW_SQRT = 5, UNIT_PHILOX = 6;

// predicates of one collision word, packed: bit 0 collision_occurred(), bit 1 collision_avoided(),
// bits 2..5 collision_occurred(CHROM_BOUNDARY / LEF_BAR / LEF_LEF_PRIMARY / LEF_LEF_SECONDARY),
// bits 6..9 collision_avoided(the same four)
MODLE_DEV u32 cw_predicates(u32 w) {
  const u32 kinds[4] = {EV_CHROM_BOUNDARY, EV_LEF_BAR, EV_LEF_LEF_PRIMARY, EV_LEF_LEF_SECONDARY};
  u32 f = (cw_occurred(w) ? 1u : 0u) | ((!cw_occurred(w) && w != 0) ? 2u : 0u);
#pragma unroll
  for (u32 k = 0; k < 4; ++k) {
    f |= cw_occurred_as(w, kinds[k]) ? (4u << k) : 0u;
    f |= cw_avoided_as(w, kinds[k]) ? (64u << k) : 0u;
  }
  return f;
}

// `in`: n pairs of 64-bit values; `out`: what the unit produces (see the cases); uniform
MODLE_DEV u32 run_test_units(const Params& p, const Interval& iv, const Workspace& ws,
                             const WaveLds& lds, u32 what, const u64* in, u32 n, u64* out) {
  const u32 lane = wave::lane();
  const Interval ivg = interval_in_device_memory(iv);
  const u64 zero[4] = {1, 2, 3, 4};
  Cell c;
  init_cell(c, p, ivg, ws, lds, n, zero);
  c.n_active = n;
  if (what == UNIT_LOOP_STATS) {
    // pairs (rev position, fwd position) of LEF i; identity ranking
    for (u32 base = 0; base < n; base += 64) {
      const u32 k = base + lane;
      if (k < n) {
        c.ws.r_pos[k] = static_cast<u32>(in[2 * k]);
        c.ws.f_pos[k] = static_cast<u32>(in[2 * k + 1]);
        c.ws.r_id[k] = k;
        c.ws.f_id[k] = k;
      }
    }
    wave::sync_mem();
    const LoopStats st = loop_size_stats(c);
    if (lane == 0) {
      out[0] = static_cast<u64>(__builtin_bit_cast(i64, st.avg));
      out[1] = static_cast<u64>(__builtin_bit_cast(i64, st.std));
    }
  } else if (what == UNIT_MATRIX_INCREMENT) {
    // pairs (row, col): ContactMatrixDense::increment
    for (u32 base = 0; base < n; base += 64) {
      const u32 k = base + lane;
      if (k < n) matrix_increment(ivg, in[2 * k], in[2 * k + 1]);
    }
  } else if (what == UNIT_COLLISION_WORDS) {
    // pairs (index, event): out = (word, predicates)
    for (u32 base = 0; base < n; base += 64) {
      const u32 k = base + lane;
      if (k < n) {
        const u32 w = cw_make(static_cast<u32>(in[2 * k]), static_cast<u32>(in[2 * k + 1]));
        out[2 * k] = (static_cast<u64>(cw_event(w)) << 56) | cw_index(w);
        out[2 * k + 1] = cw_predicates(w);
      }
    }
  } else if (what == UNIT_MATH_LOG_EXP || what == UNIT_MATH_POW_SQRT) {
    // pairs (bits of x, bits of y): out = (log x, exp y) or (pow(x, y), sqrt x), as bit images:
    // the floating-point library the path uses (wave::f_*), evaluated per lane
    for (u32 base = 0; base < n; base += 64) {
      const u32 k = base + lane;
      if (k < n) {
        const f64 x = __builtin_bit_cast(f64, in[2 * k]), y = __builtin_bit_cast(f64, in[2 * k + 1]);
        const f64 a = what == UNIT_MATH_LOG_EXP ? wave::f_log(x) : wave::f_pow(x, y);
        const f64 b = what == UNIT_MATH_LOG_EXP ? wave::f_exp(y) : wave::f_sqrt(x);
        out[2 * k] = __builtin_bit_cast(u64, a);
        out[2 * k + 1] = __builtin_bit_cast(u64, b);
      }
    }
  } else if (what == UNIT_PHILOX) {
    // pairs (counter words 0..1 | 2..3 as two 64-bit values) followed by (key words 0..1, unused):
    // two pairs per vector; out = the four output words as two 64-bit values, then zeros
    for (u32 base = 0; base < n / 2; base += 64) {
      const u32 v = base + lane;
      if (v < n / 2) {
        const u64 c_lo = in[4 * v], c_hi = in[4 * v + 1], key = in[4 * v + 2];
        u32 x[4];
        philox4x32_10(static_cast<u32>(c_lo), static_cast<u32>(c_lo >> 32), static_cast<u32>(c_hi),
                      static_cast<u32>(c_hi >> 32), static_cast<u32>(key), static_cast<u32>(key >> 32), x);
        out[4 * v] = (static_cast<u64>(x[1]) << 32) | x[0];
        out[4 * v + 1] = (static_cast<u64>(x[3]) << 32) | x[2];
        out[4 * v + 2] = 0;
        out[4 * v + 3] = 0;
      }
    }
  } else {
    return ERR_INTERNAL;
  }
  wave::sync_mem();
  return 0;
}

}  // namespace modle_dev

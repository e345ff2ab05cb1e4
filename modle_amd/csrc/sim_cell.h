// sim_cell.h (part of sim_device.h) -- one wavefront simulates one cell: the per-epoch loop of
// Simulation::simulate_one_cell (reference: src/libmodle/cpu/simulation.cpp:896-986) written for
// a 64-lane wave.  Included after a `wave` backend (wave_hip.h on the GPU).
//
// Data layout (sim_types.h: Workspace).  Extrusion units are kept in RANK ORDER, rev and fwd
// units separately: r_pos[k] / r_move[k] / r_coll[k] / r_id[k] describe the k-th rev unit in
// 5'->3' order.  Every pass that walks units in genomic order -- move adjustment, all collision
// passes, extrusion -- therefore streams contiguous memory.  The things the reference does in
// LEF-id order because of the PRNG draw order (move generation, release, bind) use id-ordered
// arrays and cross over through the unit ids, or -- for the few LEFs a phase needs -- through sparse
// entries of the inverse permutations r_rank / f_rank (see Cell::inv_valid).
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
  // upper bound of the fwd moves of this epoch (0xFFFFFFFF: unknown), set by the move adjustment
  u32 max_fwd_move;
  // ws.r_rank / ws.f_rank ([0] rev, [1] fwd) hold the complete inverse permutation.  The rank
  // update of the epoch loop does not write it (one scattered store per unit and epoch): the
  // sparse consumers -- bind, release, fix_secondary -- get the ranks of the few LEFs they need
  // from sweeps that pass over the id arrays anyway (RankFilter below), everything else
  // (contact sampling, the general rank update, the phase-level hooks) calls ensure_inverse.
  bool inv_valid[2];
  // the secondary pass collects the LEFs of its avoided collisions in the LDS id filter (for the
  // rank lookups of fix_secondary)
  bool filter_on;
  // ws.by_id_pos holds the current position of every LEF's two units (written by the burn-in
  // statistics of this epoch, behind the bind phase; positions only change again in the extrusion):
  // fix_secondary reads the partner units' positions there instead of looking their ranks up
  bool by_id_valid;
  // helper-wave mode (sim_pair.h): sequence number of the last request posted to the helper
  u32 pair_seq;
  u32 pair_interval;  // index of the task's interval (goes with every request)
  bool pair_on;       // a helper serves this epoch's requests (sampled once per epoch)
  bool ring_lent;     // the generator (and its ring in LDS) is with the helper: pair_request .. pair_take_back
#ifdef MODLE_PHASE_TIMERS
  u64 ph[16];     // profiling build: time spent per phase (wave::clock ticks)
#endif
};
// (slot 14: LEF activation; a sub-phase measurement may claim slots 14 and 15 and sends the
// activation time to slot 1 with the bind phase)
#if defined(MODLE_SUBTIMER_LEFBAR) || defined(MODLE_SUBTIMER_STATS) || defined(MODLE_SUBTIMER_RANK)
#define MODLE_SUBTIMER 1
#endif
#ifdef MODLE_SUBTIMER
#define MODLE_PH_ACTIVATION 1
#else
#define MODLE_PH_ACTIVATION 14
#endif
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
// units the one-sweep rank update (rank_update_listed) can re-insert per epoch and direction: the keys
// of the LDS sort buffer less one sentinel.  (Round 4: was STAGE_CAP = 256, which BASELINE configs[4]
// -- 64 LEFs/Mb: 250-430 LEFs released per epoch on the large chromosomes -- exceeded in every epoch.)
constexpr u32 RANK_KEY_CAP = SORT_LDS_CAP - 1;
// ... and when the update borrows the generator's ring for its keys (sim_bind_rank.h); the ring's
// contents wait at this offset (64-bit words) of ws.sort_keys, behind the scratch words of the sweeps
constexpr u32 RANK_KEY_CAP_BIG = RNG_RING - 1;
constexpr u32 RING_SPILL_AT = 64;
constexpr u32 ERR_LIST_OVERFLOW = 1;
constexpr u32 ERR_TRIAL_OVERFLOW = 2;
constexpr u32 ERR_INTERNAL = 3;
// NARROW class only: a move beyond MOVE_LIMIT (sim_types.h).  The host classes a launch NARROW only when the
// parameters rule such a move out (modle_hip_size_class); this is the net under that proof.
constexpr u32 ERR_MOVE_RANGE = 5;
constexpr bool NARROW_MOVES = sizeof(move_t) < sizeof(u32);
// (ERR_CANCELLED = 4, sim_rng.h: the host raised the abort word -- modle_hip_cancel, or the deadline
// of modle_hip_wait)

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

template <class T>
MODLE_DEV void swap_ptr(T*& a, T*& b) {
  T* t = a;
  a = b;
  b = t;
}

// an id / move array changes places with a scratch array (a 32-bit slot either way: sim_types.h)
template <class T>
MODLE_DEV void swap_with_scratch(T*& a, u32*& scratch) {
  T* t = a;
  a = reinterpret_cast<T*>(scratch);
  scratch = reinterpret_cast<u32*>(t);
}
// a scratch array seen as ids / moves
MODLE_DEV lefid_t* as_ids(u32* scratch) { return reinterpret_cast<lefid_t*>(scratch); }
MODLE_DEV const lefid_t* as_ids(const u32* scratch) { return reinterpret_cast<const lefid_t*>(scratch); }
MODLE_DEV move_t* as_moves(u32* scratch) { return reinterpret_cast<move_t*>(scratch); }
MODLE_DEV const move_t* as_moves(const u32* scratch) { return reinterpret_cast<const move_t*>(scratch); }

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
  const lefid_t* ids = FWD ? ws.f_id : ws.r_id;
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

}  // namespace modle_dev

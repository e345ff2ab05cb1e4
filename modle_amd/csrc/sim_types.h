// sim_types.h -- plain data shared by the device code (sim_device.h) and its launchers.
#pragma once
#include <stdint.h>

namespace modle_dev {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i32 = int32_t;
using i64 = int64_t;
using f64 = double;

// Positions and binding epochs are held as 32-bit values on the device (every human chromosome
// is < 2^32 bp; the host rejects longer intervals).  A released LEF has all three at UNBOUND,
// the 32-bit image of the reference's numeric_limits<bp_t>::max() marker
// (reference: src/libmodle/internal/extrusion_factors_impl.hpp:96-131).
constexpr u32 UNBOUND = 0xFFFFFFFFu;

// Collision word, 32-bit image of Collision<uint_fast32_t>
// (reference: src/libmodle/cpu/include/modle/collision_encoding.hpp:54-109): index in the low 24
// bits, the 5 event flags above them.
constexpr u32 CW_SHIFT = 24;
constexpr u32 CW_INDEX_MASK = 0x00FFFFFFu;
constexpr u32 CW_EVENT_MASK = 0x1Fu;
// device-only annotation above the event bits: the barrier of this LEF-BAR collision blocks the
// unit's own direction (a "hard" stall for release_lefs, reference: simulation.cpp:553-601)
constexpr u32 CW_HARD = 0x20000000u;
constexpr u32 EV_COLLISION = 0x10u;
constexpr u32 EV_CHROM_BOUNDARY = 0x08u;
constexpr u32 EV_LEF_BAR = 0x04u;
constexpr u32 EV_LEF_LEF_PRIMARY = 0x02u;
constexpr u32 EV_LEF_LEF_SECONDARY = 0x01u;

constexpr u32 DIR_FWD = 1u;
constexpr u32 DIR_REV = 2u;
constexpr u32 CS_NOISIFY = 1u;
constexpr u32 CS_TAD = 2u;
constexpr u32 CS_LOOP = 4u;

// PRNG block generator geometry: every lane produces RNG_CHUNK consecutive outputs per block.
// MODLE_WAVES_PER_CU (8 or 16) trades per-wave LDS and registers for resident waves.
#ifndef MODLE_WAVES_PER_CU
#define MODLE_WAVES_PER_CU 8
#endif
constexpr u32 RNG_CHUNK = MODLE_WAVES_PER_CU > 8 ? 4 : 8;
constexpr u32 RNG_BLOCK = 64 * RNG_CHUNK;      // raws per block
constexpr u32 RNG_RING = 2 * RNG_BLOCK;        // raws held in LDS per wave
// The hop to the next block costs twice what a block of 256 outputs costs to emit, so the 12-wave
// kernels (whose ring holds two blocks of 256) hop once per TWO blocks: a lane owns a run of
// RNG_RUN = 8 consecutive outputs of a super-block of 512, lanes 0-31 emit theirs into the first
// block of the pair (and every lane hops, by T^512), lanes 32-63 theirs into the second, their
// hopped states parked in the meantime (32 x 4 words behind the lane states).
constexpr u32 RNG_SPLIT = MODLE_WAVES_PER_CU > 8 ? 2 : 1;  // blocks per hop
constexpr u32 RNG_RUN = RNG_CHUNK * RNG_SPLIT;             // consecutive outputs a lane owns
constexpr u32 RNG_HOP = RNG_BLOCK * RNG_SPLIT;             // the jump table is T^RNG_HOP
constexpr u32 RNG_STATE_WORDS = 4 * 64 + (RNG_SPLIT == 2 ? 4 * 32 : 0);
constexpr u32 JUMP_TABLE_WORDS = 64 * 16 * 4;  // u64 words (32 KiB)

// Parameters of the path, digested once on the host from modle_hip_config.
struct Params {
  f64 rev_speed, fwd_speed;                // after burn-in
  f64 rev_speed_burnin, fwd_speed_burnin;
  f64 rev_std, fwd_std;
  f64 p_release, p_release_burnin;
  f64 hard_stall_mult, soft_stall_mult;
  f64 p_bypass;
  f64 pblock_major, pblock_minor;
  f64 tad_to_loop_ratio;
  f64 gev_mu, gev_sigma, gev_xi;
  f64 target_contact_density;
  u64 min_burnin_epochs, max_burnin_epochs;
  u64 burnin_target_epochs_for_lef_activation;
  u32 bin_size;
  u32 sampling_strategy;
  u32 skip_burnin;
  u32 hist_len;   // burnin_history_length
  u32 window;     // burnin_smoothing_window_size
  u32 track_1d;
#ifdef MODLE_EXP_SWITCH
  // measurement builds (make exp FLAGS=-DMODLE_EXP_SWITCH): switches read from MODLE_HIP_EXP at every launch,
  // so that ONE process compares two forms of a piece of code (bench.py: MODLE_BENCH_ALTERNATE) -- the
  // steps of a process agree to 0.1 %, two processes differ by 2 % (profiles/r04z)
  u32 exp_flags;
  u32 exp_pad_;
#endif
};

struct Interval {
  u32 start, end;  // [start, end)
  u32 n_barriers;
  u32 pad_;
  const u32* bar_pos;        // sorted ascending
  const u8* bar_dir;         // DIR_FWD / DIR_REV (blocking direction)
  const f64* bar_stp_active;
  const f64* bar_stp_inactive;
  const f64* bar_occupancy;  // compute_occupancy_from_stp(stp_active, stp_inactive)
  u32* contacts;             // band matrix, nrows*ncols+1 words
  u64* occupancy_1d;         // ncols words or nullptr
  u64* missed_updates;       // one counter
  u64 nrows, ncols;
  // bar_bucket[b] = index of the first barrier at or after start + (b << bucket_shift); lets a
  // unit find the barriers next to it with one lookup instead of a binary search
  const u32* bar_bucket;
  u32 bucket_shift;
  u32 n_buckets;
};
constexpr u32 BAR_BUCKET_SHIFT = 13;

struct Task {
  u32 interval;
  u32 num_lefs;
  u64 cell_id;
  u64 num_target_epochs;
  u64 num_target_contacts;
  u64 contacts_per_epoch;  // Simulation::compute_contacts_per_epoch(num_lefs)
  u64 prng[4];
};

struct CellResult {
  u64 epochs, burnin_epochs, num_contacts, raws_consumed;
  u64 prng_final[4];
  u64 sum_active_lefs, sampling_events, sim_epochs;
};

// Size classes (round 5: fewer bytes per unit and sweep).  The kernels exist twice.  NARROW, the class of every
// real chromosome: a launch whose largest cell has fewer than 65 536 LEFs and whose moves provably stay below
// MOVE_LIMIT (modle_hip_size_class, host_logic.cpp) keeps LEF ids and moves as 16-bit values -- the four
// rank-ordered id / move arrays and the id-ordered moves of an epoch, a third of what the sweeps of an epoch
// read and write.  WIDE (-DMODLE_WIDE): 32-bit ids and moves, for everything else.  The class is a property of
// the launch, chosen on the host; results do not depend on it (tests run both).
#ifdef MODLE_WIDE
using lefid_t = u32;
using move_t = u32;
#else
using lefid_t = u16;
using move_t = u16;
#endif
// marker left in r_move / f_move by bind: "this unit was (re)bound this epoch"
constexpr u32 NEW_MARK = static_cast<move_t>(~0u);
// ... and by the extrusion sweep: "this unit ended up below a unit of lower rank" (moves are
// bounded by the interval's length -- WIDE -- or by the class's bound -- NARROW --, below both marks)
constexpr u32 DISP_MARK = NEW_MARK - 1;
// largest move a unit can hold: a draw or an adjustment beyond it ends the cell with ERR_MOVE_RANGE (the host
// picks the WIDE class whenever the parameters allow such a move: it cannot happen in a launch it has classed
// NARROW short of a 40-sigma draw)
constexpr u32 MOVE_LIMIT = DISP_MARK - 1;

// Per-wave state in device memory (sized for the largest task of the launch).
//
// Extrusion units are stored in RANK ORDER (5'->3'), rev and fwd units separately, so that the
// passes that walk the units in genomic order (move adjustment, every collision pass, extrusion)
// read and write contiguous memory.  What the reference indexes by LEF id (binding epoch, the
// PRNG draw order of moves / release / bind) lives in id-ordered arrays, with the two inverse
// permutations r_rank / f_rank connecting the views.
constexpr u32 NUM_TMP = 10;
struct Workspace {
  u32 *r_pos, *r_coll;  // rev units, by rev rank
  u32 *f_pos, *f_coll;  // fwd units, by fwd rank
  lefid_t *r_id, *f_id;     // LEF id of the unit at a rank
  move_t *r_move, *f_move;  // move of the unit at a rank (or NEW_MARK / DISP_MARK)
  u32 *epoch, *r_rank, *f_rank, *stall; // by LEF id
  u32* tmp[NUM_TMP];                    // L words each
  // rev / fwd position of every LEF in LEF-id order, as of the last evaluation of the burn-in
  // statistics (which restores the id order anyway: sim_burnin.h); valid while Cell::by_id_valid
  u32* by_id_pos[2];
  u64* sort_keys;                       // pow2ceil(L) words
  f64* hist;                            // 2 * hist_len doubles (burn-in history)
  u8* bar_active;                       // n_barriers bytes
  // barriers that stall a unit without a Bernoulli trial (blocking probabilities in {0, 1}),
  // compacted in position order once per epoch: [0] as the rev units see them, [1] the fwd units
  u32* hit_pos[2];                      // capacity_barriers words each
  u32* hit_idx[2];                      // barrier index | hard << 31
  u32 capacity_lefs, capacity_barriers;
};
constexpr u32 NUM_STATE_ARRAYS = 12 + NUM_TMP + 2;
// (every array keeps a slot of capacity_lefs 32-bit words in both classes: a scratch array serves as positions
// in one pass and as ids or moves in the next; the NARROW class simply touches half of such a slot)

// LDS-resident (or host-emulated) per-wave context.
struct WaveLds {
  u64* ring;              // RNG_RING raws
  u64* rng_state;         // xoshiro256++ state of every lane: 4 x 64 words (word-major) [+ 4 x 32 parked: RNG_SPLIT]
  const u64* jump_table;  // T^RNG_HOP nibble table
  const f64* zig_norm_x;  // 129
  const f64* zig_norm_y;  // 129
  const f64* zig_exp_x;   // 257
  const f64* zig_exp_y;   // 257
  u64* sort_lds;          // SORT_LDS_CAP keys (ranking of newly bound units)
  u32* stage;             // STAGE_CAP words (staged slice of sorted positions)
  u64* rng_snap;          // 2 x 4 words: generator state at the start of the two blocks in the ring
  u32* mbox;              // helper-wave mode (sim_pair.h): the pair's hand-over words in LDS, or nullptr
  bool pair_dynamic;      // ... the helper may attach while the cell runs (PAIR_STATE says whether one has)
  const u32* abort_flag;  // device word polled once per epoch (cancellation) or nullptr
  u64* trace;             // optional per-epoch trace (4 words per epoch) or nullptr
  u32 trace_cap;          // epochs the trace buffer holds
  u64* phase_ticks;       // profiling build: 16 per-phase tick counters (device memory) or nullptr
  u64* state_log;         // --log-model-internal-state build: STATE_LOG_WORDS per epoch, or nullptr
  u32 state_log_cap;      // epochs the log of this task holds
};
// words of one record of the model-internal-state log (Simulation::dump_stats,
// reference: simulation.cpp:995-1056): epoch | burn-in flag << 63, barriers occupied, active LEFs,
// units stalled rev / fwd, LEFs stalled at both ends, LEF-BAR / primary / secondary collisions,
// sum of the loop sizes
constexpr u32 STATE_LOG_WORDS = 10;
constexpr u32 SORT_LDS_CAP = MODLE_WAVES_PER_CU > 8 ? 256 : 512;
constexpr u32 STAGE_CAP = 256;

}  // namespace modle_dev

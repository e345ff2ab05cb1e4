// sim_rng.h -- PRNG block generator and distributions of the device code (first half of the
// per-cell simulation; sim_device.h holds the epoch loop).
//
// One wavefront simulates one cell: the per-epoch loop of
// Simulation::simulate_one_cell (reference: src/libmodle/cpu/simulation.cpp:896-986) written for
// a 64-lane wave.  Included after a `wave` backend (wave_hip.h on the GPU).
//
// Conventions
//   * "uniform" values are identical in all 64 lanes; every collective (ballot / shuffle / sync)
//     is issued from wave-uniform control flow.
//   * Per-cell state lives in the wave's Workspace (device memory): extrusion units in rank order,
//     rev and fwd separately, plus id-ordered binding epochs and the inverse permutations
//     (layout in sim_types.h / sim_device.h); 32-bit fields throughout.
//   * The cell's single xoshiro256++ stream is produced in blocks of RNG_BLOCK raw outputs by
//     all 64 lanes (lane l owns RNG_CHUNK consecutive outputs of every block and hops to its
//     chunk of the next block with a GF(2) jump table; the 12-wave kernels hop once per pair of
//     blocks: rng_gen_block_call) and consumed strictly in the reference's order; draws whose raw-output count is data dependent are resolved with a
//     speculate / verify / restart scheme so the stream position of every draw is exact.
#pragma once
#include "sim_types.h"

namespace modle_dev {

// =============================================================================================
// small helpers
// =============================================================================================
MODLE_DEV u64 lanemask_lt(u32 lane) { return (u64(1) << lane) - 1; }
MODLE_DEV u32 cw_make(u32 idx, u32 ev) { return (idx & CW_INDEX_MASK) | (ev << CW_SHIFT); }
MODLE_DEV u32 cw_event(u32 c) { return (c >> CW_SHIFT) & CW_EVENT_MASK; }
MODLE_DEV u32 cw_index(u32 c) { return c & CW_INDEX_MASK; }
MODLE_DEV bool cw_occurred(u32 c) { return (cw_event(c) & EV_COLLISION) != 0; }
MODLE_DEV bool cw_occurred_as(u32 c, u32 what) { return cw_event(c) == (what | EV_COLLISION); }
MODLE_DEV bool cw_avoided_as(u32 c, u32 what) { return !cw_occurred(c) && cw_event(c) == what; }
MODLE_DEV u32 umin(u32 a, u32 b) { return a < b ? a : b; }
MODLE_DEV u32 umax(u32 a, u32 b) { return a > b ? a : b; }
MODLE_DEV u64 umin64(u64 a, u64 b) { return a < b ? a : b; }
MODLE_DEV i64 imin64(i64 a, i64 b) { return a < b ? a : b; }
MODLE_DEV i64 imax64(i64 a, i64 b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------------------
// 64-lane inclusive prefix scans on DPP lane moves (wave::scan_move)
// ---------------------------------------------------------------------------------------------
#define MODLE_SCAN_STEPS(APPLY)                                                               \
  APPLY(wave::SCAN_SHR1) APPLY(wave::SCAN_SHR2) APPLY(wave::SCAN_SHR4) APPLY(wave::SCAN_SHR8) \
  APPLY(wave::SCAN_BCAST15) APPLY(wave::SCAN_BCAST31)

MODLE_DEV u32 wave_prefix_max_u32(u32 v) {
#define MODLE_STEP(S) v = umax(v, wave::scan_move<S>(v, 0u));
  MODLE_SCAN_STEPS(MODLE_STEP)
#undef MODLE_STEP
  return v;
}
MODLE_DEV u32 wave_prefix_sum_u32(u32 v) {
#define MODLE_STEP(S) v += wave::scan_move<S>(v, 0u);
  MODLE_SCAN_STEPS(MODLE_STEP)
#undef MODLE_STEP
  return v;
}

// Segmented scan element: `cont` = the chain through this lane continues into the lanes before
// it.  combine(a, b), b covering the lanes before a's:  a.cont ? (pick(a.val, b.val), b.cont) : a.
// MAX = true picks the larger value, false the smaller one.
struct SegScan {
  i64 val;
  bool cont;
};
template <bool MAX>
MODLE_DEV SegScan wave_prefix_segscan(SegScan x) {
  u32 lo = static_cast<u32>(static_cast<u64>(x.val)), hi = static_cast<u32>(static_cast<u64>(x.val) >> 32);
  u32 ct = x.cont ? 1u : 0u;
  // neutral element: the other value never wins, and the chain stays open
  const u64 neutral = MAX ? 0x8000000000000000ull : 0x7FFFFFFFFFFFFFFFull;
  const u32 nlo = static_cast<u32>(neutral), nhi = static_cast<u32>(neutral >> 32);
#define MODLE_STEP(S)                                                                     \
  {                                                                                       \
    const u32 blo = wave::scan_move<S>(lo, nlo), bhi = wave::scan_move<S>(hi, nhi);        \
    const u32 bct = wave::scan_move<S>(ct, 1u);                                           \
    const i64 a = static_cast<i64>((static_cast<u64>(hi) << 32) | lo);                    \
    const i64 b = static_cast<i64>((static_cast<u64>(bhi) << 32) | blo);                  \
    const i64 r = MAX ? imax64(a, b) : imin64(a, b);                                      \
    if (ct != 0) {                                                                        \
      lo = static_cast<u32>(static_cast<u64>(r));                                         \
      hi = static_cast<u32>(static_cast<u64>(r) >> 32);                                   \
      ct = bct;                                                                           \
    }                                                                                     \
  }
  MODLE_SCAN_STEPS(MODLE_STEP)
#undef MODLE_STEP
  SegScan out;
  out.val = static_cast<i64>((static_cast<u64>(hi) << 32) | lo);
  out.cont = ct != 0;
  return out;
}

// The same on 32-bit values (half the lane moves and no 64-bit compare / select); the callers
// use it when every value of the pass fits (positions below 2^31).
template <bool MAX>
MODLE_DEV SegScan wave_prefix_segscan32(i32 value, bool cont) {
  u32 v = static_cast<u32>(value);
  u32 ct = cont ? 1u : 0u;
  const u32 neutral = MAX ? 0x80000000u : 0x7FFFFFFFu;
#define MODLE_STEP(S)                                                          \
  {                                                                            \
    const u32 bv = wave::scan_move<S>(v, neutral);                             \
    const u32 bct = wave::scan_move<S>(ct, 1u);                                \
    const i32 a = static_cast<i32>(v), b = static_cast<i32>(bv);               \
    const i32 r = MAX ? (a > b ? a : b) : (a < b ? a : b);                     \
    if (ct != 0) {                                                             \
      v = static_cast<u32>(r);                                                 \
      ct = bct;                                                                \
    }                                                                          \
  }
  MODLE_SCAN_STEPS(MODLE_STEP)
#undef MODLE_STEP
  SegScan out;
  out.val = static_cast<i64>(static_cast<i32>(v));
  out.cont = ct != 0;
  return out;
}

constexpr f64 TWO64 = 18446744073709551616.0;
constexpr f64 TWO_M64 = 5.42101086242752217e-20;
constexpr f64 TWO_M56 = 1.387778780781445675529539585113525390625e-17;
constexpr f64 DBL_EPS = 2.220446049250313e-16;

// =============================================================================================
// PRNG: xoshiro256++ block generator (reference stream: random.hpp:26-32)
// =============================================================================================
struct Rng {
  u64* ring;           // RNG_RING raws (LDS)
  const u64* jump;     // T^RNG_BLOCK nibble table (LDS)
  u64* state;          // per lane xoshiro state at the start of its chunk of the NEXT block:
                       // word w of lane l at state[w * 64 + l] (LDS)
  u64* snap;           // generator state at the start of each of the two blocks in the ring: word w
                       // of the block with parity b at snap[4 * b + w] (LDS)
  u64 gen_end;         // uniform: raws [gen_end - RNG_RING, gen_end) are in the ring
  u64 pos;             // uniform: stream position of the next raw to be consumed
  u32* feed;           // helper-wave mode: hand-over words of the wave that produces the blocks for
                       // this consumer (rng_take_fed_block), or nullptr: the consumer produces them
  const u32* feed_abort;  // fed stream: the host's abort word, read while waiting for a block (or nullptr)
  u32 feed_error;      // fed stream: non-zero once a wait for the producer was abandoned (the outputs
                       // consumed since are not the stream's: the consumer reports the status)
};
// Hand-over words between a consumer of the stream and the wave that produces its blocks (u32, LDS;
// helper-wave mode, sim_helper.h).  The producer replaces the older block of the ring once the
// consumer's published position is FEED_MARGIN outputs into the newer one (a consumer hands back
// up to 66 outputs at the end of a direction of generate_moves: they must still be in the ring).
constexpr u32 FEED_START = 0;    // consumer -> producer: sequence number of the session
constexpr u32 FEED_STOP = 1;     // consumer -> producer: the session ends
constexpr u32 FEED_ACK = 2;      // producer -> consumer: it has (no block in the making)
constexpr u32 FEED_POS = 3;      // consumer -> producer: low word of its position
constexpr u32 FEED_GEN_END = 4;  // low word of the end of the ring (consumer at the start, then the producer)
constexpr u32 FEED_EXIT = 5;     // consumer -> producer, with a new session number: leave
constexpr u32 FEED_MARGIN = 128;

// Every spin loop of the hand-over protocols (this file: a consumer waiting for its producer;
// sim_pair.h / sim_helper.h: main wave, helper, producer) is bounded by the host: a waiting wave reads
// the abort word (host memory, one round trip over the fabric) every SPIN_POLL naps and leaves its
// loop when the host has raised it -- modle_hip_cancel, or modle_hip_wait once its deadline has
// passed.  A wave whose partner has stopped answering (a protocol bug) therefore costs the launch, not
// the box.  (The reference polls `_ctx` once per epoch, simulation.cpp:933; its workers never wait
// for each other: scheduler_simulate.cpp:264-270.)
constexpr u32 SPIN_POLL = 1024;
constexpr u32 ERR_CANCELLED = 4;  // the host raised the abort word (reference: _ctx polled per epoch)
// one step of a spin loop: naps, and every SPIN_POLL steps says whether the host has raised the abort
// word (`spins`: the loop's step counter; `abort_flag` may be nullptr: never aborted)
MODLE_DEV bool spin_nap_aborted(const u32* abort_flag, u32& spins) {
  wave::nap();
  if ((++spins & (SPIN_POLL - 1)) != 0 || abort_flag == nullptr) return false;
  return wave::uniform(wave::load_system_u32(abort_flag)) != 0;
}

MODLE_DEV u64 rotl64(u64 x, int k) { return (x << k) | (x >> (64 - k)); }

MODLE_DEV u64 xo_next(u64& s0, u64& s1, u64& s2, u64& s3) {
  const u64 result = rotl64(s0 + s3, 23) + s0;
  const u64 t = s1 << 17;
  s2 ^= s0;
  s3 ^= s1;
  s1 ^= s2;
  s0 ^= s3;
  s2 ^= t;
  s3 = rotl64(s3, 45);
  return result;
}

#ifdef MODLE_RNG_PHILOX
constexpr u32 RNG_SWZ = RNG_CHUNK;  // (counter based: every lane fills its chunk of every block)
#else
constexpr u32 RNG_SWZ = RNG_RUN;
#endif
MODLE_DEV u32 ring_index(u64 p) {
  const u32 off = static_cast<u32>(p) & (RNG_BLOCK - 1);
  const u32 blk = (static_cast<u32>(p) / RNG_BLOCK) & 1u;
  // run-local XOR swizzle: lanes writing element t of their runs hit distinct LDS banks
  return blk * RNG_BLOCK + (off ^ ((off / RNG_SWZ) & (RNG_SWZ - 1)));
}

// T^RNG_HOP * state: XOR of one table row (4 words) per state nibble.  Rows are fetched in groups
// of four (all loads of a group in flight, then folded; left alone the compiler waits for every LDS
// load before issuing the next one), each row as two 128-bit reads, and folded on 32-bit halves with
// three-input XORs.
MODLE_DEV void rng_hop(const MODLE_LDS u64* jump, const u64 w[4], u64 j[4]) {
  u32 acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int GROUP = 4;
#pragma unroll
  for (int g = 0; g < 64 / GROUP; ++g) {
    wave::LdsRow r[GROUP];
#pragma unroll
    for (int q = 0; q < GROUP; ++q) {
      const int nib = g * GROUP + q;  // nibble k of state word wi
      const int wi = nib / 16, k = nib % 16;
      const u32 half = k < 8 ? static_cast<u32>(w[wi]) : static_cast<u32>(w[wi] >> 32);
      const u32 v = (half >> (4 * (k % 8))) & 15u;
      r[q] = wave::lds_load_row(jump + ((wi * 16 + k) * 16) * 4, v);
    }
    wave::sched_fence();
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      acc[h] = wave::xor3(acc[h], r[0].h[h], r[1].h[h]);
      acc[h] = wave::xor3(acc[h], r[2].h[h], r[3].h[h]);
      wave::pin(acc[h]);  // fold this group before the next group's rows are fetched
    }
    wave::sched_fence();
  }
  j[0] = (static_cast<u64>(acc[1]) << 32) | acc[0];
  j[1] = (static_cast<u64>(acc[3]) << 32) | acc[2];
  j[2] = (static_cast<u64>(acc[5]) << 32) | acc[4];
  j[3] = (static_cast<u64>(acc[7]) << 32) | acc[6];
}

// Produces one block of the stream into the ring half `ring_base`.  A real call: it is reached from
// every phase that draws, and its registers stay out of the callers' allocation.
//   RNG_SPLIT == 1 (8-wave kernels): every lane emits its RNG_CHUNK outputs and hops to its chunk
//     of the next block (state <- T^RNG_BLOCK * state through the nibble table).
//   RNG_SPLIT == 2 (12-wave kernels): blocks come in pairs (ring half 0, then ring half 1; the
//     callers alternate, starting with half 0).  Half 0: lanes 0-31 emit their runs of RNG_RUN
//     outputs, and EVERY lane hops from the start of its run by T^RNG_HOP -- lanes 0-31 keep the
//     result as their state, lanes 32-63 park it (they have not emitted yet).  Half 1: lanes 32-63
//     emit their runs and take the parked states.  One hop per 512 outputs instead of two.
MODLE_DEV_CALL void rng_gen_block_call(MODLE_LDS u64* ring, const MODLE_LDS u64* jump,
                                       MODLE_LDS u64* state, MODLE_LDS u64* snap, u32 ring_base) {
  const u32 lane = wave::lane();
  u64 a0 = state[0 * 64 + lane], a1 = state[1 * 64 + lane], a2 = state[2 * 64 + lane],
      a3 = state[3 * 64 + lane];
  const u64 w[4] = {a0, a1, a2, a3};
  if constexpr (RNG_SPLIT == 2) {
    const bool second = wave::uniform(ring_base) != 0;
    MODLE_LDS u64* park = state + 4 * 64;
    if (lane == (second ? 32u : 0u)) {
      // this lane sits at the first output of the block: the engine state a sequential generator
      // would have there (rng_final_state recovers the state at any position inside the ring)
      MODLE_LDS u64* sn = snap + (second ? 4 : 0);
      sn[0] = a0;
      sn[1] = a1;
      sn[2] = a2;
      sn[3] = a3;
    }
    if ((lane >= 32) == second) {
      const u32 base = ring_base + RNG_RUN * (lane & 31u);
#pragma unroll
      for (u32 t = 0; t < RNG_RUN; ++t) {
        ring[base + (t ^ (lane & (RNG_RUN - 1)))] = xo_next(a0, a1, a2, a3);
      }
    }
    if (second) {
      if (lane >= 32) {
#pragma unroll
        for (u32 k = 0; k < 4; ++k) state[k * 64 + lane] = park[k * 32 + (lane - 32)];
      }
      return;
    }
    u64 j[4];
    rng_hop(jump, w, j);
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
      if (lane < 32) {
        state[k * 64 + lane] = j[k];
      } else {
        park[k * 32 + (lane - 32)] = j[k];
      }
    }
    return;
  }
  if (lane == 0) {
    // lane 0 sits at the first output of the block
    MODLE_LDS u64* sn = snap + 4 * (ring_base / RNG_BLOCK);
    sn[0] = a0;
    sn[1] = a1;
    sn[2] = a2;
    sn[3] = a3;
  }
  const u32 base = ring_base + RNG_CHUNK * lane;
#pragma unroll
  for (u32 t = 0; t < RNG_CHUNK; ++t) {
    ring[base + (t ^ (lane & (RNG_CHUNK - 1)))] = xo_next(a0, a1, a2, a3);
  }
  u64 j[4];
  rng_hop(jump, w, j);
  state[0 * 64 + lane] = j[0];
  state[1 * 64 + lane] = j[1];
  state[2 * 64 + lane] = j[2];
  state[3 * 64 + lane] = j[3];
}

// Philox4x32-10 (the round function of the PHILOX generator policy below; compiled into every
// build so that the known-answer vectors run against the default library too)
MODLE_DEV void philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1, u32 out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const u64 p0 = static_cast<u64>(0xD2511F53u) * c0;
    const u64 p1 = static_cast<u64>(0xCD9E8D57u) * c2;
    const u32 n0 = static_cast<u32>(p1 >> 32) ^ c1 ^ k0;
    const u32 n1 = static_cast<u32>(p1);
    const u32 n2 = static_cast<u32>(p0 >> 32) ^ c3 ^ k1;
    const u32 n3 = static_cast<u32>(p0);
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

#ifndef MODLE_RNG_PHILOX
// consumer side of a fed stream: publishes the position and waits for the producer's next block
MODLE_DEV void rng_take_fed_block(Rng& g) {
  u32* f = g.feed;
  wave::st_release_wg(&f[FEED_POS], static_cast<u32>(g.pos));
  u32 spins = 0;
  for (;;) {
    const u32 fed = wave::uniform(wave::ld_acquire_wg(&f[FEED_GEN_END]));
    const u32 ahead = fed - static_cast<u32>(g.gen_end);  // (the producer is less than 2^31 outputs ahead)
    if (ahead != 0) {
      g.gen_end = wave::known_uniform(g.gen_end + ahead);
      return;
    }
    if (g.feed_error != 0 || spin_nap_aborted(g.feed_abort, spins)) {
      // abandoned: the consumer goes on over whatever the ring holds (every loop that draws ends
      // with probability one on any data), nothing it computes from here on is used
      g.feed_error = ERR_CANCELLED;
      g.gen_end = wave::known_uniform(g.gen_end + RNG_BLOCK);
      return;
    }
  }
}
MODLE_DEV void rng_gen_block(Rng& g) {
  if (g.feed != nullptr) {
    rng_take_fed_block(g);
    return;
  }
  wave::lockstep();  // other lanes may still be reading the block that is about to be replaced
  rng_gen_block_call((MODLE_LDS u64*)g.ring, (const MODLE_LDS u64*)g.jump, (MODLE_LDS u64*)g.state,
                     (MODLE_LDS u64*)g.snap, ((static_cast<u32>(g.gen_end) / RNG_BLOCK) & 1u) * RNG_BLOCK);
  g.gen_end += RNG_BLOCK;
  wave::sync_lds();
}

MODLE_DEV void rng_init(Rng& g, const u64 state[4]) {
  const u32 lane = wave::lane();
  u64 s0 = state[0], s1 = state[1], s2 = state[2], s3 = state[3];
  // lane l starts at its run (RNG_SPLIT == 1: chunk) of the first block (pair of blocks)
  for (u32 k = 0; k < RNG_RUN * 63; ++k) {
    if (k < RNG_RUN * lane) (void)xo_next(s0, s1, s2, s3);
  }
  wave::lockstep();
  g.state[0 * 64 + lane] = s0;
  g.state[1 * 64 + lane] = s1;
  g.state[2 * 64 + lane] = s2;
  g.state[3 * 64 + lane] = s3;
  g.gen_end = 0;
  g.pos = 0;
  g.feed = nullptr;
  g.feed_abort = nullptr;
  g.feed_error = 0;
  wave::sync_lds();
}
#else
// ---------------------------------------------------------------------------------------------
// PHILOX generator policy (compile-time, -DMODLE_RNG_PHILOX; SURVEY.md H1's second back-end).
// The cell's stream is counter based: output p of the stream is one half of
// Philox4x32-10(counter = (p >> 1, c2, c3), key = (k0, k1)), the 128 bits (k0, k1, c2, c3) taken
// from the task's PRNG state words (so cells keep distinct streams through the same per-cell
// jump() as in the exact mode).  Every lane computes its outputs directly -- no state, no GF(2)
// jump table -- and everything downstream (the ring, the draw order, speculate / verify /
// replay) is unchanged.  Results are NOT comparable bit for bit with the reference's xoshiro
// stream: this mode is validated statistically (modle_amd/evaluate.py) and bit for bit only
// against the oracle running the same policy.  Round function and constants: Salmon et al.,
// "Parallel random numbers: as easy as 1, 2, 3" (SC'11), as implemented by rocRAND's
// philox4x32_10.
// ---------------------------------------------------------------------------------------------
// one block of the stream; a real call like the xoshiro block generator (it is reached from every
// phase that draws)
MODLE_DEV_CALL void rng_philox_block_call(MODLE_LDS u64* ring, u64 key, u64 hi, u64 block_start,
                                          u32 ring_base) {
  const u32 lane = wave::lane();
  const u32 base = ring_base + RNG_CHUNK * lane;
  const u64 q0 = (block_start + RNG_CHUNK * lane) >> 1;
#pragma unroll
  for (u32 t = 0; t < RNG_CHUNK / 2; ++t) {
    const u64 q = q0 + t;
    u32 x[4];
    philox4x32_10(static_cast<u32>(q), static_cast<u32>(q >> 32), static_cast<u32>(hi),
                  static_cast<u32>(hi >> 32), static_cast<u32>(key), static_cast<u32>(key >> 32), x);
    ring[base + ((2 * t) ^ (lane & (RNG_CHUNK - 1)))] = (static_cast<u64>(x[1]) << 32) | x[0];
    ring[base + ((2 * t + 1) ^ (lane & (RNG_CHUNK - 1)))] = (static_cast<u64>(x[3]) << 32) | x[2];
  }
}

MODLE_DEV void rng_gen_block(Rng& g) {
  wave::lockstep();  // other lanes may still be reading the block that is about to be replaced
  // the stream's identity lives in g.snap[4..5] (LDS): key and the upper counter words
  const u64 key = wave::uniform(g.snap[4]), hi = wave::uniform(g.snap[5]);
  rng_philox_block_call((MODLE_LDS u64*)g.ring, key, hi, g.gen_end,
                        ((static_cast<u32>(g.gen_end) / RNG_BLOCK) & 1u) * RNG_BLOCK);
  g.gen_end += RNG_BLOCK;
  wave::sync_lds();
}

MODLE_DEV void rng_init(Rng& g, const u64 state[4]) {
  wave::lockstep();
  if (wave::lane() == 0) {
    g.snap[0] = state[0];
    g.snap[1] = state[1];
    g.snap[2] = state[2];
    g.snap[3] = state[3];
    g.snap[4] = state[0] ^ state[2];
    g.snap[5] = state[1] ^ state[3];
  }
  g.gen_end = 0;
  g.pos = 0;
  g.feed = nullptr;
  g.feed_abort = nullptr;
  g.feed_error = 0;
  wave::sync_lds();
}
#endif

// State of the sequential engine after g.pos outputs (what the reference's PRNG object holds when
// the cell returns).  The block that contains g.pos is one of the two in the ring (g.pos >
// g.gen_end - RNG_RING: a block is only generated when a consumer needs outputs beyond
// gen_end), or the one that has not been generated yet.  Once per cell; uniform.
MODLE_DEV void rng_final_state(const Rng& g, u64 out[4]) {
#ifdef MODLE_RNG_PHILOX
  // counter based: the engine state is the task's seed state, the position is raws_consumed
  for (int w = 0; w < 4; ++w) out[w] = wave::uniform(g.snap[w]);
  return;
#endif
  const u64 q = g.pos / RNG_BLOCK;
  u64 s0, s1, s2, s3;
  if (q * RNG_BLOCK == g.gen_end) {
    // next block: lane 0's chunk starts there (the second block of a pair: lane 32's run)
    const u32 first = (RNG_SPLIT == 2 && (q & 1u) != 0) ? 32u : 0u;
    s0 = wave::uniform(g.state[0 * 64 + first]);
    s1 = wave::uniform(g.state[1 * 64 + first]);
    s2 = wave::uniform(g.state[2 * 64 + first]);
    s3 = wave::uniform(g.state[3 * 64 + first]);
  } else {
    const u32 b = static_cast<u32>(q & 1u);
    s0 = wave::uniform(g.snap[4 * b + 0]);
    s1 = wave::uniform(g.snap[4 * b + 1]);
    s2 = wave::uniform(g.snap[4 * b + 2]);
    s3 = wave::uniform(g.snap[4 * b + 3]);
  }
  const u32 r = static_cast<u32>(g.pos - q * RNG_BLOCK);
  for (u32 k = 0; k < r; ++k) (void)xo_next(s0, s1, s2, s3);
  out[0] = s0;
  out[1] = s1;
  out[2] = s2;
  out[3] = s3;
}

// makes raws [pos, pos + k) readable (k <= RNG_BLOCK); uniform
MODLE_DEV void rng_ensure(Rng& g, u32 k) {
  // (both are the same in every lane; saying so keeps them in scalar registers and the loop a
  // scalar branch)
  g.pos = wave::known_uniform(g.pos);
  g.gen_end = wave::known_uniform(g.gen_end);
  while (g.gen_end < g.pos + k) rng_gen_block(g);
}
// consumes n raws (uniform)
MODLE_DEV void rng_advance(Rng& g, u64 n) { g.pos = wave::known_uniform(g.pos + n); }
MODLE_DEV u64 rng_peek(const Rng& g, u64 p) { return g.ring[ring_index(p)]; }
// uniform: next raw of the stream
MODLE_DEV u64 rng_next(Rng& g) {
  rng_ensure(g, 1);
  return wave::uniform(rng_peek(g, g.pos++));
}
// =============================================================================================
// Distributions (Boost.Random 1.88 semantics on a 64-bit engine; reference aliases:
// src/common/include/modle/common/random.hpp:34-53).  "exact" routines are executed uniformly by
// the whole wave and consume the stream sequentially; "fast" forms evaluate one speculative draw
// per lane from a raw output that has already been fetched.
// =============================================================================================
MODLE_DEV bool bernoulli_raw(u64 raw, f64 p) { return static_cast<f64>(raw) <= p * TWO64; }
MODLE_DEV f64 canonical_raw(u64 raw) {
  f64 r = static_cast<f64>(raw) / TWO64;
  if (r == 1.0) r -= DBL_EPS / 2;
  return r;
}
MODLE_DEV f64 uniform01_exact(Rng& g) {
  for (;;) {
    const f64 r = static_cast<f64>(rng_next(g)) * TWO_M64;
    if (r < 1.0) return r;
  }
}
MODLE_DEV u64 uniform_int_bucket(u64 range) {
  u64 bucket = ~u64(0) / (range + 1);
  if (~u64(0) % (range + 1) == range) ++bucket;
  return bucket;
}
// floor(raw / d) for a wave-uniform divisor, without the (long) 64-bit division sequence: a
// double-precision product gives the quotient to within one, the remainder tells which way.
// `inv` = 1.0 / (f64)d.  Exact while d <= 2^62 and the quotient stays below 2^40 (relative error
// of the product < 2^-50), which holds for position draws: d = bucket >= 2^32.
MODLE_DEV u64 udiv_by_uniform(u64 raw, u64 d, f64 inv) {
  u64 q = static_cast<u64>(static_cast<f64>(raw) * inv);
  const u64 rem = raw - q * d;  // modulo 2^64
  if (static_cast<i64>(rem) < 0) {
    --q;  // candidate one too large: rem in [-d, 0)
  } else if (rem >= d) {
    ++q;  // one too small: rem in [d, 2 d)
  }
  return q;
}

// uniform_int_distribution<u64>{0, range}, range != 0 and != 2^64-1
MODLE_DEV u64 uniform_int_exact(Rng& g, u64 range, u64 bucket) {
  for (;;) {
    const u64 r = rng_next(g) / bucket;
    if (r <= range) return r;
  }
}

MODLE_DEV f64 int_float_pair8(u64 raw, u32& bucket) {
  bucket = static_cast<u32>(raw) & 0xFFu;
  const u64 u = raw & ~((u64(1) << 11) - 1);
  return static_cast<f64>(u >> 8) * TWO_M56;
}

MODLE_DEV f64 unit_exponential_exact(Rng& g, const WaveLds& lds) {
  f64 shift = 0.0;
  for (;;) {
    u32 i;
    const f64 u = int_float_pair8(rng_next(g), i);
    const f64 x = u * lds.zig_exp_x[i];
    if (x < lds.zig_exp_x[i + 1]) return shift + x;
    if (i == 0) {
      shift += lds.zig_exp_x[1];
    } else {
      const f64 y01 = uniform01_exact(g);
      const f64 y = lds.zig_exp_y[i] + y01 * (lds.zig_exp_y[i + 1] - lds.zig_exp_y[i]);
      const f64 y_above_ubound =
          (lds.zig_exp_x[i] - lds.zig_exp_x[i + 1]) * y01 - (lds.zig_exp_x[i] - x);
      const f64 y_above_lbound =
          y - (lds.zig_exp_y[i + 1] + (lds.zig_exp_x[i + 1] - x) * lds.zig_exp_y[i + 1]);
      if (y_above_ubound < 0 && (y_above_lbound < 0 || y < wave::f_exp(-x))) return x + shift;
    }
  }
}

// boost unit_normal_distribution; uniform, consumes from g.pos
MODLE_DEV_NOINLINE f64 unit_normal_exact(Rng& g, const WaveLds& lds) {
  for (;;) {
    u32 b;
    const f64 u = int_float_pair8(rng_next(g), b);
    const f64 sign = (b & 1u) ? 1.0 : -1.0;
    const u32 i = b >> 1;
    const f64 x = u * lds.zig_norm_x[i];
    if (x < lds.zig_norm_x[i + 1]) return x * sign;
    if (i == 0) {
      const f64 tail_start = lds.zig_norm_x[1];
      for (;;) {
        const f64 tx = unit_exponential_exact(g, lds) / tail_start;
        const f64 ty = unit_exponential_exact(g, lds);
        if (2 * ty > tx * tx) return (tx + tail_start) * sign;
      }
    }
    const f64 y01 = uniform01_exact(g);
    const f64 xi = lds.zig_norm_x[i], xi1 = lds.zig_norm_x[i + 1];
    const f64 yi = lds.zig_norm_y[i], yi1 = lds.zig_norm_y[i + 1];
    const f64 y = yi + y01 * (yi1 - yi);
    const f64 chord = (xi - xi1) * y01 - (xi - x);
    const f64 tangent = y - (yi + (xi - x) * yi * xi);
    const f64 y_above_ubound = (xi >= 1) ? chord : tangent;
    const f64 y_above_lbound = (xi >= 1) ? tangent : chord;
    if (y_above_ubound < 0 && (y_above_lbound < 0 || y < wave::f_exp(-(x * x / 2)))) {
      return x * sign;
    }
  }
}

// boost poisson_distribution<size_t, double>; uniform
MODLE_DEV_NOINLINE u64 poisson_exact(Rng& g, f64 mean) {
  if (mean < 10) {
    f64 p = wave::f_exp(-mean);
    u64 x = 0;
    f64 u = uniform01_exact(g);
    while (u > p) {
      u = u - p;
      ++x;
      p = mean * p / static_cast<f64>(x);
    }
    return x;
  }
  const f64 log_fact[10] = {0.0,
                            0.0,
                            0.69314718055994529,
                            1.7917594692280550,
                            3.1780538303479458,
                            4.7874917427820458,
                            6.5792512120101012,
                            8.5251613610654147,
                            10.604602902745251,
                            12.801827480081469};
  const f64 smu = wave::f_sqrt(mean);
  const f64 b = 0.931 + 2.53 * smu;
  const f64 a = -0.059 + 0.02483 * b;
  const f64 inv_alpha = 1.1239 + 1.1328 / (b - 3.4);
  const f64 v_r = 0.9277 - 3.6224 / (b - 2);
  for (;;) {
    f64 u;
    f64 v = uniform01_exact(g);
    if (v <= 0.86 * v_r) {
      u = v / v_r - 0.43;
      return static_cast<u64>(
          wave::f_floor((2 * a / (0.5 - wave::f_abs(u)) + b) * u + mean + 0.445));
    }
    if (v >= v_r) {
      u = uniform01_exact(g) - 0.5;
    } else {
      u = v / v_r - 0.93;
      u = ((u < 0) ? -0.5 : 0.5) - u;
      v = uniform01_exact(g) * v_r;
    }
    const f64 us = 0.5 - wave::f_abs(u);
    if (us < 0.013 && v > us) continue;
    const f64 k = wave::f_floor((2 * a / us + b) * u + mean + 0.445);
    v = v * inv_alpha / (a / (us * us) + b);
    const f64 log_sqrt_2pi = 0.91893853320467267;
    if (k >= 10) {
      if (wave::f_log(v * smu) <= (k + 0.5) * wave::f_log(mean / k) - mean - log_sqrt_2pi + k -
                                      (1 / 12. - (1 / 360. - 1 / (1260. * k * k)) / (k * k)) / k) {
        return static_cast<u64>(k);
      }
    } else if (k >= 0) {
      f64 lf = 0.0;
      const int ki = static_cast<int>(k);
#pragma unroll
      for (int t = 0; t < 10; ++t) lf = (t == ki) ? log_fact[t] : lf;
      if (wave::f_log(v) <= k * wave::f_log(mean) - mean - lf) return static_cast<u64>(k);
    }
  }
}

MODLE_DEV f64 binom_fc(i64 k) {
  const f64 table[10] = {0.08106146679532726, 0.04134069595540929, 0.02767792568499834,
                         0.02079067210376509, 0.01664469118982119, 0.01387612882307075,
                         0.01189670994589177, 0.01041126526197209, 0.009255462182712733,
                         0.008330563433362871};
  if (k < 10) {
    f64 v = 0.0;
#pragma unroll
    for (int t = 0; t < 10; ++t) v = (t == static_cast<int>(k)) ? table[t] : v;
    return v;
  }
  const f64 ikp1 = 1.0 / static_cast<f64>(k + 1);
  return (1.0 / 12 - (1.0 / 360 - (1.0 / 1260) * (ikp1 * ikp1)) * (ikp1 * ikp1)) * ikp1;
}

// boost binomial_distribution<ptrdiff_t, double>{t, p}; uniform
MODLE_DEV_NOINLINE i64 binomial_exact(Rng& g, i64 t, f64 p_) {
  const f64 p = (0.5 < p_) ? (1 - p_) : p_;
  const i64 m = static_cast<i64>(static_cast<f64>(t + 1) * p);
  i64 k;
  if (m < 11) {
    const f64 q = 1 - p;
    const f64 s = p / q;
    const f64 a = static_cast<f64>(t + 1) * s;
    f64 r = wave::f_pow(1 - p, static_cast<f64>(t));
    f64 u = uniform01_exact(g);
    k = 0;
    while (u > r) {
      u = u - r;
      ++k;
      const f64 r1 = ((a / static_cast<f64>(k)) - s) * r;
      if (r1 < DBL_EPS && r1 < r) break;
      r = r1;
    }
    return (0.5 < p_) ? t - k : k;
  }
  const f64 r = p / (1 - p);
  const f64 nr = static_cast<f64>(t + 1) * r;
  const f64 npq = static_cast<f64>(t) * p * (1 - p);
  const f64 sqrt_npq = wave::f_sqrt(npq);
  const f64 b = 1.15 + 2.53 * sqrt_npq;
  const f64 a = -0.0873 + 0.0248 * b + 0.01 * p;
  const f64 c = static_cast<f64>(t) * p + 0.5;
  const f64 alpha = (2.83 + 5.1 / b) * sqrt_npq;
  const f64 v_r = 0.92 - 4.2 / b;
  const f64 u_rv_r = 0.86 * v_r;
  for (;;) {
    f64 u;
    f64 v = uniform01_exact(g);
    if (v <= u_rv_r) {
      u = v / v_r - 0.43;
      k = static_cast<i64>(wave::f_floor((2 * a / (0.5 - wave::f_abs(u)) + b) * u + c));
      break;
    }
    if (v >= v_r) {
      u = uniform01_exact(g) - 0.5;
    } else {
      u = v / v_r - 0.93;
      u = ((u < 0) ? -0.5 : 0.5) - u;
      v = uniform01_exact(g) * v_r;
    }
    const f64 us = 0.5 - wave::f_abs(u);
    k = static_cast<i64>(wave::f_floor((2 * a / us + b) * u + c));
    if (k < 0 || k > t) continue;
    v = v * alpha / (a / (us * us) + b);
    const i64 kmi = k > m ? k - m : m - k;
    const f64 km = static_cast<f64>(kmi);
    if (km <= 15) {
      f64 f = 1;
      if (m < k) {
        i64 i = m;
        do {
          ++i;
          f = f * (nr / static_cast<f64>(i) - r);
        } while (i != k);
      } else if (m > k) {
        i64 i = k;
        do {
          ++i;
          v = v * (nr / static_cast<f64>(i) - r);
        } while (i != m);
      }
      if (v <= f) break;
      continue;
    }
    v = wave::f_log(v);
    const f64 rho = (km / npq) * (((km / 3. + 0.625) * km + 1. / 6) / npq + 0.5);
    const f64 tt = -km * km / (2 * npq);
    if (v < tt - rho) break;
    if (v > tt + rho) continue;
    const i64 nm = t - m + 1;
    const f64 h = (static_cast<f64>(m) + 0.5) *
                      wave::f_log(static_cast<f64>(m + 1) / (r * static_cast<f64>(nm))) +
                  binom_fc(m) + binom_fc(t - m);
    const i64 nk = t - k + 1;
    if (v <= h +
                 static_cast<f64>(t + 1) *
                     wave::f_log(static_cast<f64>(nm) / static_cast<f64>(nk)) +
                 (static_cast<f64>(k) + 0.5) *
                     wave::f_log(static_cast<f64>(nk) * r / static_cast<f64>(k + 1)) -
                 binom_fc(k) - binom_fc(t - k)) {
      break;
    }
  }
  return (0.5 < p_) ? t - k : k;
}

// genextreme_value_distribution (reference: genextreme_value_distribution.hpp:87-105)
MODLE_DEV f64 genextreme_from_canonical(f64 u, f64 mu, f64 sigma, f64 xi) {
  if (xi == 0.0) return (mu - sigma) * wave::f_log(-wave::f_log(u));
  return mu + (sigma * (1.0 - wave::f_pow(-wave::f_log(u), xi))) / xi;
}


}  // namespace modle_dev

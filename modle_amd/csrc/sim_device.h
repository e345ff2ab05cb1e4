// sim_device.h -- one wavefront simulates one cell: the per-epoch loop of
// Simulation::simulate_one_cell (reference: src/libmodle/cpu/simulation.cpp:896-986) written for
// a 64-lane wave.  Included after a `wave` backend (wave_hip.h on the GPU).
//
// Conventions
//   * "uniform" values are identical in all 64 lanes; every collective (ballot / shuffle / sync)
//     is issued from wave-uniform control flow.
//   * Per-cell state lives in the wave's Workspace (device memory), indexed by LEF id with two
//     rank arrays giving the 5'->3' order of rev / fwd units, like the reference's State buffers
//     (reference: src/libmodle/cpu/include/modle/simulation.hpp:86-94) but with 32-bit fields.
//   * The cell's single xoshiro256++ stream is produced in blocks of RNG_BLOCK raw outputs by
//     all 64 lanes (lane l owns RNG_CHUNK consecutive outputs of every block and hops to its
//     chunk of the next block with a GF(2) jump table) and consumed strictly in the reference's
//     order; draws whose raw-output count is data dependent are resolved with a
//     speculate / verify / restart scheme so the stream position of every draw is exact.
#pragma once
#include "sim_types.h"

namespace modle_dev {

// =============================================================================================
// small helpers
// =============================================================================================
MODLE_DEV u64 lanemask_lt(u32 lane) { return (u64(1) << lane) - 1; }
MODLE_DEV u32 cw_make(u32 idx, u32 ev) { return (idx & CW_INDEX_MASK) | (ev << CW_SHIFT); }
MODLE_DEV u32 cw_event(u32 c) { return c >> CW_SHIFT; }
MODLE_DEV u32 cw_index(u32 c) { return c & CW_INDEX_MASK; }
MODLE_DEV bool cw_occurred(u32 c) { return (cw_event(c) & EV_COLLISION) != 0; }
MODLE_DEV bool cw_occurred_as(u32 c, u32 what) { return cw_event(c) == (what | EV_COLLISION); }
MODLE_DEV bool cw_avoided_as(u32 c, u32 what) { return !cw_occurred(c) && cw_event(c) == what; }
MODLE_DEV u32 umin(u32 a, u32 b) { return a < b ? a : b; }
MODLE_DEV u32 umax(u32 a, u32 b) { return a > b ? a : b; }
MODLE_DEV u64 umin64(u64 a, u64 b) { return a < b ? a : b; }
MODLE_DEV i64 imin64(i64 a, i64 b) { return a < b ? a : b; }
MODLE_DEV i64 imax64(i64 a, i64 b) { return a > b ? a : b; }

constexpr f64 TWO64 = 18446744073709551616.0;
constexpr f64 TWO_M64 = 5.42101086242752217e-20;
constexpr f64 TWO_M56 = 1.387778780781445675529539585113525390625e-17;
constexpr f64 DBL_EPS = 2.220446049250313e-16;

// =============================================================================================
// PRNG: xoshiro256++ block generator (reference stream: random.hpp:26-32)
// =============================================================================================
struct Rng {
  u64 s0, s1, s2, s3;  // per lane: state at the start of this lane's chunk of the NEXT block
  u64* ring;           // RNG_RING raws (LDS)
  const u64* jump;     // T^RNG_BLOCK nibble table (LDS)
  u64 gen_end;         // uniform: raws [gen_end - RNG_RING, gen_end) are in the ring
  u64 pos;             // uniform: stream position of the next raw to be consumed
};

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

MODLE_DEV u32 ring_index(u64 p) {
  const u32 off = static_cast<u32>(p) & (RNG_BLOCK - 1);
  const u32 blk = (static_cast<u32>(p) / RNG_BLOCK) & 1u;
  // chunk-local XOR swizzle: lanes writing element t of their chunks hit distinct LDS banks
  return blk * RNG_BLOCK + (off ^ ((off / RNG_CHUNK) & (RNG_CHUNK - 1)));
}

MODLE_DEV void rng_jump(Rng& g) {
  u64 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  const u64 w[4] = {g.s0, g.s1, g.s2, g.s3};
#pragma unroll
  for (int wi = 0; wi < 4; ++wi) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const u32 v = static_cast<u32>(w[wi] >> (4 * k)) & 15u;
      const u64* row = g.jump + ((wi * 16 + k) * 16 + v) * 4;
      a0 ^= row[0];
      a1 ^= row[1];
      a2 ^= row[2];
      a3 ^= row[3];
    }
  }
  g.s0 = a0;
  g.s1 = a1;
  g.s2 = a2;
  g.s3 = a3;
}

MODLE_DEV void rng_gen_block(Rng& g) {
  wave::lockstep();  // other lanes may still be reading the block that is about to be replaced
  const u32 lane = wave::lane();
  u64 a0 = g.s0, a1 = g.s1, a2 = g.s2, a3 = g.s3;
  const u32 base = ((static_cast<u32>(g.gen_end) / RNG_BLOCK) & 1u) * RNG_BLOCK + RNG_CHUNK * lane;
#pragma unroll
  for (u32 t = 0; t < RNG_CHUNK; ++t) {
    g.ring[base + (t ^ (lane & (RNG_CHUNK - 1)))] = xo_next(a0, a1, a2, a3);
  }
  rng_jump(g);
  g.gen_end += RNG_BLOCK;
  wave::sync_mem();
}

MODLE_DEV void rng_init(Rng& g, const u64 state[4]) {
  const u32 lane = wave::lane();
  g.s0 = state[0];
  g.s1 = state[1];
  g.s2 = state[2];
  g.s3 = state[3];
  // lane l starts RNG_CHUNK * l outputs into the stream
  for (u32 k = 0; k < RNG_CHUNK * 63; ++k) {
    if (k < RNG_CHUNK * lane) (void)xo_next(g.s0, g.s1, g.s2, g.s3);
  }
  g.gen_end = 0;
  g.pos = 0;
}

// makes raws [pos, pos + k) readable (k <= RNG_BLOCK); uniform
MODLE_DEV void rng_ensure(Rng& g, u32 k) {
  while (g.gen_end < g.pos + k) rng_gen_block(g);
}
MODLE_DEV u64 rng_peek(const Rng& g, u64 p) { return g.ring[ring_index(p)]; }
// uniform: next raw of the stream
MODLE_DEV u64 rng_next(Rng& g) {
  rng_ensure(g, 1);
  return rng_peek(g, g.pos++);
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
MODLE_DEV f64 unit_normal_exact(Rng& g, const WaveLds& lds) {
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
MODLE_DEV u64 poisson_exact(Rng& g, f64 mean) {
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
MODLE_DEV i64 binomial_exact(Rng& g, i64 t, f64 p_) {
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

// =============================================================================================
// Cell context
// =============================================================================================
struct Cell {
  const Params* p;
  const Interval* iv;
  Workspace ws;
  WaveLds lds;
  Rng g;
  u32 n_lefs;    // Task::num_lefs
  u32 n_active;  // State::num_active_lefs
  u32 hist_len;  // entries in the burn-in history buffers
  u32 hist_head; // ring head
  u32 error;     // non-zero when an internal capacity was exceeded (uniform)
};
constexpr u32 ERR_LIST_OVERFLOW = 1;
constexpr u32 ERR_TRIAL_OVERFLOW = 2;

// =============================================================================================
// select_and_bind_lefs (reference: simulation.cpp:988-993, simulation_impl.hpp:30-91)
// =============================================================================================
MODLE_DEV void phase_bind(Cell& c, u32 epoch_now) {
  const Interval& iv = *c.iv;
  Workspace& ws = c.ws;
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  const u64 range = static_cast<u64>(iv.end) - 1 - iv.start;
  const u64 bucket = range != 0 ? uniform_int_bucket(range) : 1;
  for (u32 base = 0; base < n; base += 64) {
    const u32 i = base + lane;
    const bool unb = i < n && ws.epoch[i] == UNBOUND;
    const u64 mask = wave::ballot(unb);
    if (mask == 0) continue;
    u32 posv = iv.start;
    if (range != 0) {
      const u32 cnt = static_cast<u32>(wave::popc64(mask));
      rng_ensure(c.g, cnt);
      const u32 k = static_cast<u32>(wave::popc64(mask & lanemask_lt(lane)));
      const u64 r = rng_peek(c.g, c.g.pos + k) / bucket;
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
        c.g.pos += cnt;
      }
    }
    if (unb) {
      ws.rev_pos[i] = posv;
      ws.fwd_pos[i] = posv;
      ws.epoch[i] = epoch_now;
    }
  }
  wave::sync_mem();
}

// =============================================================================================
// rank_lefs (reference: simulation.cpp:410-496)
//
// Total order: position, then binding epoch (rev: older first, fwd: younger first), then the
// position in the incoming rank array (the reference leaves this last tie to an unstable sort;
// see DESIGN.md "ranking ties").  Units that were already ranked stay sorted across an epoch, so
// the update is: split the rank array into carried-over and newly bound units, sort the new
// ones, merge, then repair epoch ties.
// =============================================================================================
MODLE_DEV u32 pow2_ceil(u32 x) {
  u32 p = 1;
  while (p < x) p <<= 1;
  return p;
}

MODLE_DEV void bitonic_sort_u64(u64* keys, u32 m_pow2) {
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
      wave::sync_mem();
    }
  }
}

// full comparator: position, binding epoch (rev: older first, fwd: younger first), position in
// the incoming rank array (`where`, by LEF id)
template <bool FWD>
MODLE_DEV bool rank_pair_out_of_order(const Workspace& ws, const u32* pos, const u32* where, u32 a,
                                      u32 b) {
  const u32 pa = pos[a], pb = pos[b];
  if (pa != pb) return pa > pb;
  const u32 ea = ws.epoch[a], eb = ws.epoch[b];
  if (ea != eb) return FWD ? ea < eb : ea > eb;
  return where[a] > where[b];
}

// all_new: treat every entry as newly bound (full sort; used by the phase-level test entry point)
template <bool FWD>
MODLE_DEV void rank_update(Cell& c, u32 epoch_now, bool all_new) {
  Workspace& ws = c.ws;
  const u32 n = c.n_active;
  if (n < 2) return;
  const u32 lane = wave::lane();
  u32* rank = FWD ? ws.fwd_rank : ws.rev_rank;
  const u32* pos = FWD ? ws.fwd_pos : ws.rev_pos;
  u32* old_ids = ws.tmp_a;
  u32* old_pos = ws.tmp_b;
  u32* new_ids = ws.tmp_c;
  u32* where = ws.tmp_d;
  u64* keys = ws.sort_keys;

  // 1. stable split.  A carried-over unit that is no longer in order (its position is below the
  //    running maximum of the carried-over units before it; this can happen after
  //    fix_secondary_lef_lef_collisions re-positions a pair) is handled like a new unit, so that
  //    the kept sequence is non-decreasing by construction.
  u32 n_old = 0, n_new = 0;
  u32 run_max = 0;  // max position of carried-over units in previous batches
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    const bool act = k < n;
    const u32 id = act ? rank[k] : 0;
    const u32 P = act ? pos[id] : 0;
    const bool fresh = act && (all_new || ws.epoch[id] == epoch_now);
    const bool carried = act && !fresh;
    // exclusive prefix maximum of the carried-over positions
    u32 pm = carried ? P : 0;
#pragma unroll
    for (u32 s = 1; s < 64; s <<= 1) {
      const u32 o = wave::shfl_up(pm, s);
      if (lane >= s) pm = umax(pm, o);
    }
    const u32 incl_last = wave::bcast(pm, 63);
    const u32 pm_prev = wave::shfl_up(pm, 1);
    const u32 excl = umax(run_max, lane > 0 ? pm_prev : 0);
    const bool displaced = carried && P < excl;
    const bool is_new = fresh || displaced;
    const bool is_old = carried && !displaced;
    const u64 mn = wave::ballot(is_new);
    const u64 mo = wave::ballot(is_old);
    if (act) where[id] = k;
    if (is_new) {
      const u32 j = n_new + static_cast<u32>(wave::popc64(mn & lanemask_lt(lane)));
      new_ids[j] = id;
      keys[j] = (static_cast<u64>(P) << 32) | j;
    }
    if (is_old) {
      const u32 j = n_old + static_cast<u32>(wave::popc64(mo & lanemask_lt(lane)));
      old_ids[j] = id;
      old_pos[j] = P;
    }
    n_new += static_cast<u32>(wave::popc64(mn));
    n_old += static_cast<u32>(wave::popc64(mo));
    run_max = umax(run_max, incl_last);
  }
  wave::sync_mem();
  if (n_new != 0) {
    // 2. sort the new units by (position, previous rank)
    const u32 m2 = pow2_ceil(n_new);
    for (u32 base = n_new; base < m2; base += 64) {
      const u32 k = base + lane;
      if (k < m2) keys[k] = ~u64(0);
    }
    wave::sync_mem();
    if (m2 > 1) bitonic_sort_u64(keys, m2);
    // 3. merge by cross-ranking (kept units are sorted; equal positions are ordered later)
    for (u32 base = 0; base < n_old; base += 64) {
      const u32 a = base + lane;
      if (a < n_old) {
        const u32 p = old_pos[a];
        const u64 thr = FWD ? ((static_cast<u64>(p) + 1) << 32) : (static_cast<u64>(p) << 32);
        u32 lo = 0, hi = n_new;
        while (lo < hi) {
          const u32 mid = (lo + hi) >> 1;
          if (keys[mid] < thr) lo = mid + 1; else hi = mid;
        }
        rank[a + lo] = old_ids[a];
      }
    }
    for (u32 base = 0; base < n_new; base += 64) {
      const u32 b = base + lane;
      if (b < n_new) {
        const u64 key = keys[b];
        const u32 p = static_cast<u32>(key >> 32);
        u32 lo = 0, hi = n_old;
        while (lo < hi) {
          const u32 mid = (lo + hi) >> 1;
          const u32 q = old_pos[mid];
          const bool before = FWD ? (q < p) : (q <= p);
          if (before) lo = mid + 1; else hi = mid;
        }
        rank[b + lo] = new_ids[static_cast<u32>(key)];
      }
    }
    wave::sync_mem();
  }
  // 4. order equal positions (epoch rule, then previous rank) with a stable odd-even
  //    transposition; normally nothing moves
  bool bad = false;
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    const bool chk = k >= 1 && k < n;
    const bool b = chk && rank_pair_out_of_order<FWD>(ws, pos, where, rank[k - 1], rank[k]);
    bad = wave::any(b) || bad;
  }
  while (bad) {
    bad = false;
    for (u32 parity = 0; parity < 2; ++parity) {
      for (u32 base = 0; base < n; base += 128) {
        const u32 k = base + 2 * lane + parity;
        bool sw = false;
        if (k + 1 < n) {
          const u32 a = rank[k], b = rank[k + 1];
          if (rank_pair_out_of_order<FWD>(ws, pos, where, a, b)) {
            rank[k] = b;
            rank[k + 1] = a;
            sw = true;
          }
        }
        bad = wave::any(sw) || bad;
      }
      wave::sync_mem();
    }
  }
}

// =============================================================================================
// generate_moves (reference: simulation.cpp:272-330)
// =============================================================================================
MODLE_DEV u32 move_from_normal(f64 unit, f64 speed, f64 std) {
  const f64 v = unit * std + speed;
  return static_cast<u32>(static_cast<u64>(wave::f_round(v > 0.0 ? v : 0.0)));
}

MODLE_DEV void generate_moves_dir(Cell& c, u32* moves, f64 speed, f64 std) {
  Workspace& ws = c.ws;
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  if (std == 0.0) {
    const u32 move_int = static_cast<u32>(static_cast<u64>(wave::f_round(speed)));
    for (u32 base = 0; base < n; base += 64) {
      const u32 i = base + lane;
      if (i < n) moves[i] = ws.epoch[i] != UNBOUND ? move_int : 0;
    }
    return;
  }
  u32 i0 = 0;
  while (i0 < n) {
    const u32 cntb = umin(64u, n - i0);
    const u32 i = i0 + lane;
    const bool act = lane < cntb;
    const bool bnd = act && ws.epoch[i] != UNBOUND;
    const u64 bm = wave::ballot(bnd);
    const u32 ndraw = static_cast<u32>(wave::popc64(bm));
    rng_ensure(c.g, ndraw);
    const u32 k = static_cast<u32>(wave::popc64(bm & lanemask_lt(lane)));
    u32 bucket;
    const f64 u = int_float_pair8(rng_peek(c.g, c.g.pos + k), bucket);
    const u32 layer = bucket >> 1;
    const f64 x = u * c.lds.zig_norm_x[layer];
    const bool fast = x < c.lds.zig_norm_x[layer + 1];
    f64 unit = (bucket & 1u) ? x : -x;
    const u64 slow = wave::ballot(bnd && !fast);
    if (slow == 0) {
      if (act) moves[i] = bnd ? move_from_normal(unit, speed, std) : 0;
      c.g.pos += ndraw;
      i0 += cntb;
    } else {
      // lanes before the first slow draw are final; the slow one is replayed exactly
      const u32 f = static_cast<u32>(wave::ctz64(slow));
      c.g.pos += static_cast<u32>(wave::popc64(bm & lanemask_lt(f)));
      const f64 exact = unit_normal_exact(c.g, c.lds);
      if (lane == f) unit = exact;
      if (act && lane <= f) moves[i] = bnd ? move_from_normal(unit, speed, std) : 0;
      i0 += f + 1;
    }
  }
}

// =============================================================================================
// adjust_moves_of_consecutive_extr_units (reference: simulation.cpp:350-407) as two segmented
// scans over rank order, plus clamp_moves (reference: simulation.cpp:332-347).
//
// rev units, ranks high -> low:  land'[k] = min(land[k], land'[k+1] - 1) while both units are
// bound and neither reaches the 5'-end.  With d[k] = land[k] - k this is a segmented suffix
// minimum of d.  The reference tests "unit k+1 reaches the 5'-end" on the *updated* move of
// k+1; the scan uses the original move and the (rare, chromosome-end only) cases where the
// update changes the answer are replayed sequentially from the first affected rank.
// =============================================================================================
MODLE_DEV void adjust_moves_rev(Cell& c, const u32* mv_in, u32* mv_out) {
  Workspace& ws = c.ws;
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  const u64 start = c.iv->start;
  const u32 nbatch = (n + 63) / 64;
  i64 carry_d = 0;
  bool carry_ok = false, carry_cross = false;
  i64 viol_rank = -1;
  for (u32 bi = nbatch; bi-- > 0;) {
    const u32 k = bi * 64 + lane;
    const bool act = k < n;
    const u32 id = act ? ws.rev_rank[k] : 0;
    const u32 P = act ? ws.rev_pos[id] : 0;
    const u32 M = act ? mv_in[id] : 0;
    const bool bnd = act && ws.epoch[id] != UNBOUND;
    const bool okself = bnd && static_cast<u64>(P) > start + M;
    const i64 d = okself ? static_cast<i64>(P - M) - static_cast<i64>(k) : 0;
    const bool ok_next_in = wave::shfl_down(okself, 1);
    const bool ok_next = lane < 63 ? ok_next_in : carry_ok;
    const bool link = okself && ok_next;
    i64 val = d;
    bool cont = link;
#pragma unroll
    for (u32 s = 1; s < 64; s <<= 1) {
      const i64 ov = wave::shfl_down(val, s);
      const bool oc = wave::shfl_down(cont, s);
      if (lane + s < 64 && cont) {
        val = imin64(val, ov);
        cont = oc;
      }
    }
    if (cont) val = imin64(val, carry_d);
    u32 Mnew = M;
    if (okself) Mnew = P - static_cast<u32>(val + static_cast<i64>(k));
    if (act) mv_out[id] = Mnew;
    const bool cross = okself && static_cast<u64>(P) <= start + Mnew;
    const bool cross_next_in = wave::shfl_down(cross, 1);
    const bool cross_next = lane < 63 ? cross_next_in : carry_cross;
    const u64 vm = wave::ballot(link && cross_next);
    if (vm != 0 && viol_rank < 0) viol_rank = bi * 64 + (63 - wave::clz64(vm)) + 1;
    carry_d = wave::bcast(val, 0);
    carry_ok = wave::bcast(okself, 0);
    carry_cross = wave::bcast(cross, 0);
  }
  wave::sync_mem();
  if (viol_rank >= 0) {
    // sequential replay (reference loop) from the first rank whose decision the scan got wrong
    for (u32 i = static_cast<u32>(viol_rank); i > 0; --i) {
      const u32 i1 = ws.rev_rank[i - 1], i2 = ws.rev_rank[i];
      u32 M1 = mv_in[i1];
      if (ws.epoch[i1] != UNBOUND && ws.epoch[i2] != UNBOUND) {
        const u32 M2 = mv_out[i2];
        const u64 P1 = ws.rev_pos[i1], P2 = ws.rev_pos[i2];
        if (!(P1 <= start + M1 || P2 <= start + M2)) {
          const u64 pos1 = P1 - M1, pos2 = P2 - M2;
          if (pos2 <= pos1) M1 += static_cast<u32>(pos1 - pos2) + 1;
        }
      }
      mv_out[i1] = M1;
    }
    wave::sync_mem();
  }
}

MODLE_DEV void adjust_moves_fwd(Cell& c, const u32* mv_in, u32* mv_out) {
  Workspace& ws = c.ws;
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  const u64 last = static_cast<u64>(c.iv->end) - 1;
  const u32 nbatch = (n + 63) / 64;
  i64 carry_d = 0;
  bool carry_ok = false, carry_cross = false;
  i64 viol_rank = -1;
  for (u32 bi = 0; bi < nbatch; ++bi) {
    const u32 k = bi * 64 + lane;
    const bool act = k < n;
    const u32 id = act ? ws.fwd_rank[k] : 0;
    const u32 P = act ? ws.fwd_pos[id] : 0;
    const u32 M = act ? mv_in[id] : 0;
    const bool bnd = act && ws.epoch[id] != UNBOUND;
    const bool okself = bnd && static_cast<u64>(P) + M <= last;
    const i64 d = okself ? static_cast<i64>(static_cast<u64>(P) + M) - static_cast<i64>(k) : 0;
    const bool ok_prev_in = wave::shfl_up(okself, 1);
    const bool ok_prev = lane > 0 ? ok_prev_in : carry_ok;
    const bool link = okself && ok_prev;  // link between k-1 and k
    i64 val = d;
    bool cont = link;
#pragma unroll
    for (u32 s = 1; s < 64; s <<= 1) {
      const i64 ov = wave::shfl_up(val, s);
      const bool oc = wave::shfl_up(cont, s);
      if (lane >= s && cont) {
        val = imax64(val, ov);
        cont = oc;
      }
    }
    if (cont) val = imax64(val, carry_d);
    u32 Mnew = M;
    if (okself) Mnew = static_cast<u32>(val + static_cast<i64>(k) - static_cast<i64>(P));
    if (act) mv_out[id] = Mnew;
    const bool cross = okself && static_cast<u64>(P) + Mnew > last;
    const bool cross_prev_in = wave::shfl_up(cross, 1);
    const bool cross_prev = lane > 0 ? cross_prev_in : carry_cross;
    const u64 vm = wave::ballot(link && cross_prev);
    // lowest rank k-1 whose updated move crosses the 3'-end while the scan linked it to k
    if (vm != 0 && viol_rank < 0) viol_rank = static_cast<i64>(bi) * 64 + wave::ctz64(vm) - 1;
    carry_d = wave::bcast(val, 63);
    carry_ok = wave::bcast(okself, 63);
    carry_cross = wave::bcast(cross, 63);
  }
  wave::sync_mem();
  if (viol_rank >= 0) {
    for (u32 i = static_cast<u32>(viol_rank) + 1; i < n; ++i) {
      const u32 i1 = ws.fwd_rank[i - 1], i2 = ws.fwd_rank[i];
      u32 M2 = mv_in[i2];
      if (ws.epoch[i1] != UNBOUND && ws.epoch[i2] != UNBOUND) {
        const u32 M1 = mv_out[i1];
        const u64 P1 = ws.fwd_pos[i1], P2 = ws.fwd_pos[i2];
        if (!(P1 + M1 > last || P2 + M2 > last)) {
          const u64 pos1 = P1 + M1, pos2 = P2 + M2;
          if (pos1 >= pos2) M2 += static_cast<u32>(pos1 - pos2) + 1;
        }
      }
      mv_out[i2] = M2;
    }
    wave::sync_mem();
  }
}

MODLE_DEV void clamp_moves(Cell& c, const u32* rev_in, const u32* fwd_in) {
  Workspace& ws = c.ws;
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  for (u32 base = 0; base < n; base += 64) {
    const u32 i = base + lane;
    if (i < n) {
      u32 rm = rev_in[i], fm = fwd_in[i];
      if (ws.epoch[i] != UNBOUND) {
        rm = umin(rm, ws.rev_pos[i] - c.iv->start);
        fm = umin(fm, c.iv->end - ws.fwd_pos[i] - 1);
      }
      ws.rev_moves[i] = rm;
      ws.fwd_moves[i] = fm;
    }
  }
  wave::sync_mem();
}

MODLE_DEV void phase_generate_moves(Cell& c, bool burnin_completed) {
  const Params& p = *c.p;
  generate_moves_dir(c, c.ws.rev_moves, burnin_completed ? p.rev_speed : p.rev_speed_burnin,
                     p.rev_std);
  generate_moves_dir(c, c.ws.fwd_moves, burnin_completed ? p.fwd_speed : p.fwd_speed_burnin,
                     p.fwd_std);
  wave::sync_mem();
  adjust_moves_rev(c, c.ws.rev_moves, c.ws.tmp_a);
  adjust_moves_fwd(c, c.ws.fwd_moves, c.ws.tmp_b);
  clamp_moves(c, c.ws.tmp_a, c.ws.tmp_b);
}

// =============================================================================================
// ExtrusionBarriers::init_states / next_state (reference: extrusion_barriers.cpp:145-161,
// 219-230)
// =============================================================================================
MODLE_DEV void barriers_init_states(Cell& c) {
  const Interval& iv = *c.iv;
  const u32 nb = iv.n_barriers;
  const u32 lane = wave::lane();
  for (u32 base = 0; base < nb; base += 64) {
    const u32 i = base + lane;
    const bool act = i < nb;
    const f64 occ = act ? iv.bar_occupancy[i] : 0.0;
    const bool draws = act && occ != 0.0;  // bernoulli(0) consumes nothing
    const u64 dm = wave::ballot(draws);
    const u32 cnt = static_cast<u32>(wave::popc64(dm));
    rng_ensure(c.g, cnt);
    const u32 k = static_cast<u32>(wave::popc64(dm & lanemask_lt(lane)));
    const bool on = draws && bernoulli_raw(rng_peek(c.g, c.g.pos + k), occ);
    if (act) c.ws.bar_active[i] = on ? 1 : 0;
    c.g.pos += cnt;
  }
  wave::sync_mem();
}

MODLE_DEV void barriers_next_state(Cell& c) {
  const Interval& iv = *c.iv;
  const u32 nb = iv.n_barriers;
  const u32 lane = wave::lane();
  for (u32 base = 0; base < nb; base += 64) {
    const u32 i = base + lane;
    const u32 cnt = umin(64u, nb - base);
    rng_ensure(c.g, cnt);
    if (i < nb) {
      const f64 u = canonical_raw(rng_peek(c.g, c.g.pos + lane));
      const u8 st = c.ws.bar_active[i];
      if (!st && u > iv.bar_stp_inactive[i]) {
        c.ws.bar_active[i] = 1;
      } else if (st && u > iv.bar_stp_active[i]) {
        c.ws.bar_active[i] = 0;
      }
    }
    c.g.pos += cnt;
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
MODLE_DEV BoundaryCounts detect_boundaries(Cell& c) {
  Workspace& ws = c.ws;
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  const u32 start = c.iv->start, last = c.iv->end - 1;
  const u32 first_fwd_pos = ws.fwd_pos[ws.fwd_rank[0]];
  // position of the last bound unit in rev rank order
  u32 last_rev_pos = 0;
  for (u32 top = n; top > 0;) {
    const u32 cnt = umin(64u, top);
    const u32 k = top - 1 - lane;  // descending
    const bool act = lane < cnt;
    const u32 id = act ? ws.rev_rank[k] : 0;
    const bool bnd = act && ws.epoch[id] != UNBOUND;
    const u32 P = bnd ? ws.rev_pos[id] : 0;
    const u64 m = wave::ballot(bnd);
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
    const u32 id = act ? ws.rev_rank[k] : 0;
    const u32 P = act ? ws.rev_pos[id] : 0;
    const u32 M = act ? ws.rev_moves[id] : 0;
    const bool at = act && P == start;
    const bool brk_b = act && !at && P > first_fwd_pos;
    const bool brk_c = act && !at && !brk_b && P - M == start;
    const u64 stop = wave::ballot(brk_b || brk_c);
    const u32 s = stop != 0 ? static_cast<u32>(wave::ctz64(stop)) : 64u;
    const bool mark = act && ((lane < s && at) || (lane == s && brk_c));
    if (mark) ws.rev_coll[id] = mark5;
    out.n5 += static_cast<u32>(wave::popc64(wave::ballot(mark)));
    if (stop != 0) break;
  }
  // fwd units: ranks n-1 down to 1 (rank 0 is never visited, simulation_detect_collisions.cpp:91)
  for (u32 top = n; top > 1;) {
    const u32 cnt = umin(64u, top - 1);
    const u32 k = top - 1 - lane;
    const bool act = lane < cnt;
    const u32 id = act ? ws.fwd_rank[k] : 0;
    const bool bnd = act && ws.epoch[id] != UNBOUND;
    const u32 P = act ? ws.fwd_pos[id] : 0;
    const u32 M = act ? ws.fwd_moves[id] : 0;
    const bool unb = act && !bnd;
    const bool at = bnd && P == last;
    const bool brk_b = bnd && !at && P < last_rev_pos;
    const bool brk_c = bnd && !at && !brk_b && P + M == last;
    const u64 stop = wave::ballot(brk_b || brk_c);
    const u32 s = stop != 0 ? static_cast<u32>(wave::ctz64(stop)) : 64u;
    const bool mark = (lane < s && at) || (lane == s && brk_c);
    if (mark) ws.fwd_coll[id] = mark3;
    out.n3 += static_cast<u32>(wave::popc64(wave::ballot(mark || (lane < s && unb))));
    if (stop != 0) break;
    top -= cnt;
  }
  wave::sync_mem();
  return out;
}

MODLE_DEV u32 lower_bound_u32(const u32* a, u32 n, u32 key) {  // first index with a[i] >= key
  u32 lo = 0, hi = n;
  while (lo < hi) {
    const u32 mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// detect_lef_bar_collisions (reference: simulation_detect_collisions.cpp:123-247), evaluated
// per extrusion unit: barrier b is tested against the first rev unit downstream of it (first fwd
// unit upstream), so the barriers that can stall the unit of rank j are those between the unit
// of rank j-1 and itself that lie within its move.  Bernoulli trials (pblock not in {0,1}) are
// numbered in the reference's order: barriers ascending for rev units, descending for fwd units.
template <bool FWD>
MODLE_DEV void detect_lef_bar(Cell& c, BoundaryCounts bc) {
  Workspace& ws = c.ws;
  const Interval& iv = *c.iv;
  const Params& p = *c.p;
  const u32 n = c.n_active;
  const u32 nb = iv.n_barriers;
  if (nb == 0) return;
  const u32 lane = wave::lane();
  const u32* rank = FWD ? ws.fwd_rank : ws.rev_rank;
  const u32* pos = FWD ? ws.fwd_pos : ws.rev_pos;
  const u32* moves = FWD ? ws.fwd_moves : ws.rev_moves;
  u32* coll = FWD ? ws.fwd_coll : ws.rev_coll;
  const u32 major_dir = FWD ? DIR_FWD : DIR_REV;
  const bool trials = !((p.pblock_major == 1.0 || p.pblock_major == 0.0) &&
                        (p.pblock_minor == 1.0 || p.pblock_minor == 0.0));
  // first / last rank that takes part
  const u32 j_rev0 = bc.n5 == 0 ? 0 : bc.n5 - 1;
  const u32 j_fwd0 = bc.n3 == 0 ? n - 1 : n - bc.n3;
  u32 carry_pos = 0;  // position of the neighbouring unit processed by the previous batch
  const u32 nbatch = (n + 63) / 64;
  for (u32 bi = 0; bi < nbatch; ++bi) {
    // rev: ranks ascending; fwd: ranks descending, lane 0 = highest rank of the batch
    const i64 kk = FWD ? static_cast<i64>(j_fwd0) - static_cast<i64>(bi) * 64 - lane
                       : static_cast<i64>(j_rev0) + static_cast<i64>(bi) * 64 + lane;
    const bool act = kk >= 0 && kk < static_cast<i64>(n);
    if (!wave::any(act)) break;
    const u32 k = act ? static_cast<u32>(kk) : 0;
    const u32 id = act ? rank[k] : 0;
    const u32 P = act ? pos[id] : 0;
    const u32 M = act ? moves[id] : 0;
    const bool bnd = act && ws.epoch[id] != UNBOUND;
    // neighbour towards which the barriers are shadowed (rank k-1 for rev, k+1 for fwd)
    const u32 nbr_in = wave::shfl_up(P, 1);
    const bool first = (bi == 0 && lane == 0);
    const u32 nbr = lane > 0 ? nbr_in : carry_pos;
    // window of barrier indices [b_lo, b_hi)
    u32 b_lo = 0, b_hi = 0;
    if (bnd) {
      if (!FWD) {
        // prev <= bpos < P and P - bpos <= M
        const u32 reach = P - M;  // M <= P - start after clamping
        const u32 lo_pos = first ? reach : umax(reach, nbr);
        b_lo = lower_bound_u32(iv.bar_pos, nb, lo_pos);
        b_hi = lower_bound_u32(iv.bar_pos, nb, P);
      } else {
        // P < bpos <= next and bpos - P <= M
        const u64 reach = static_cast<u64>(P) + M;
        const u64 hi_pos = first ? reach : umin64(reach, nbr);
        b_lo = lower_bound_u32(iv.bar_pos, nb, P + 1);
        b_hi = hi_pos >= 0xFFFFFFFFull ? nb
                                        : lower_bound_u32(iv.bar_pos, nb,
                                                          static_cast<u32>(hi_pos) + 1);
      }
    }
    // number of Bernoulli trials this unit consumes
    u32 ntr = 0;
    if (trials) {
      for (u32 b = b_lo; b < b_hi; ++b) {
        const f64 pb = iv.bar_dir[b] == major_dir ? p.pblock_major : p.pblock_minor;
        ntr += (ws.bar_active[b] && pb != 1.0 && pb != 0.0) ? 1u : 0u;
      }
    }
    // exclusive prefix sum of ntr over lanes
    u32 off = ntr;
#pragma unroll
    for (u32 s = 1; s < 64; s <<= 1) {
      const u32 o = wave::shfl_up(off, s);
      if (lane >= s) off += o;
    }
    const u32 total = wave::bcast(off, 63);
    off -= ntr;
    if (total > RNG_BLOCK) {
      c.error = ERR_TRIAL_OVERFLOW;  // more Bernoulli trials in one batch than the ring holds
      return;
    }
    if (total != 0) rng_ensure(c.g, total);
    u32 winner = 0xFFFFFFFFu;
    u32 t = 0;
    for (u32 q = b_lo; q < b_hi; ++q) {
      const u32 b = FWD ? (b_hi - 1 - (q - b_lo)) : q;  // reference visiting order
      if (!ws.bar_active[b]) continue;
      const f64 pb = iv.bar_dir[b] == major_dir ? p.pblock_major : p.pblock_minor;
      bool hit;
      if (pb == 1.0) {
        hit = true;
      } else if (pb == 0.0) {
        hit = false;
      } else {
        hit = bernoulli_raw(rng_peek(c.g, c.g.pos + off + t), pb);
        ++t;
      }
      if (hit) winner = b;  // later visits overwrite earlier ones
    }
    if (winner != 0xFFFFFFFFu) coll[id] = cw_make(winner, EV_COLLISION | EV_LEF_BAR);
    c.g.pos += total;
    carry_pos = wave::bcast(P, 63);
  }
  wave::sync_mem();
}

// compute_lef_lef_collision_pos (reference: simulation.cpp:523-551)
MODLE_DEV void lef_lef_collision_pos(u32 rev_p, u32 fwd_p, u32 rev_move, u32 fwd_move,
                                     u32& out_rev, u32& out_fwd) {
  const u64 relative_speed = static_cast<u64>(rev_move) + fwd_move;
  const f64 ttc = static_cast<f64>(static_cast<u64>(rev_p - fwd_p)) / static_cast<f64>(relative_speed);
  const u32 cpos = fwd_p + static_cast<u32>(static_cast<u64>(
                               wave::f_round(static_cast<f64>(static_cast<u64>(fwd_move)) * ttc)));
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
MODLE_DEV void detect_primary(Cell& c, BoundaryCounts bc, const u32* fwd_sorted) {
  Workspace& ws = c.ws;
  const Params& p = *c.p;
  const u32 n = c.n_active;
  if (bc.n5 == n || bc.n3 == n) return;
  const u32 lane = wave::lane();
  const u32 i2 = bc.n3 == 0 ? n : n - (bc.n3 - 1);
  const bool trials = p.p_bypass != 0.0;
  const f64 p_collide = 1.0 - p.p_bypass;
  const u32 prim = EV_COLLISION | EV_LEF_LEF_PRIMARY;
  u32 carry_pos = 0;
  for (u32 base = bc.n5; base < n; base += 64) {
    const u32 k = base + lane;
    const bool act = k < n;
    const u32 rev_idx = act ? ws.rev_rank[k] : 0;
    const u32 R = act ? ws.rev_pos[rev_idx] : 0;
    const u32 prev_in = wave::shfl_up(R, 1);
    const u32 Rprev = lane > 0 ? prev_in : carry_pos;
    bool cand = false;
    u32 fwd_idx = 0, F = 0, rev_move = 0, fwd_move = 0;
    if (act) {
      const u32 pf = lower_bound_u32(fwd_sorted, n, R);  // fwd units strictly upstream of R
      if (pf >= 1 && pf < i2) {
        F = fwd_sorted[pf - 1];
        const bool first_after = (k == bc.n5) || Rprev <= F;
        if (first_after) {
          fwd_idx = ws.fwd_rank[pf - 1];
          rev_move = ws.rev_moves[rev_idx];
          fwd_move = ws.fwd_moves[fwd_idx];
          const u32 delta = R - F;  // > 0 by construction
          cand = static_cast<u64>(delta) < static_cast<u64>(rev_move) + fwd_move;
        }
      }
    }
    const u64 cm = wave::ballot(cand);
    bool hit = cand;
    if (trials && cm != 0) {
      const u32 cnt = static_cast<u32>(wave::popc64(cm));
      rng_ensure(c.g, cnt);
      const u32 t = static_cast<u32>(wave::popc64(cm & lanemask_lt(lane)));
      hit = cand && bernoulli_raw(rng_peek(c.g, c.g.pos + t), p_collide);
      c.g.pos += cnt;
    }
    if (hit) {
      u32 cpos_rev, cpos_fwd;
      lef_lef_collision_pos(R, F, rev_move, fwd_move, cpos_rev, cpos_fwd);
      const u32 rc = ws.rev_coll[rev_idx], fc = ws.fwd_coll[fwd_idx];
      const bool rev_occ = cw_occurred(rc), fwd_occ = cw_occurred(fc);
      if (!rev_occ && !fwd_occ) {
        ws.rev_coll[rev_idx] = cw_make(fwd_idx, prim);
        ws.fwd_coll[fwd_idx] = cw_make(rev_idx, prim);
      } else if (rev_occ && !fwd_occ) {
        const u32 barrier_pos = stalling_barrier_pos(*c.iv, rc);
        if (cpos_fwd > barrier_pos) ws.rev_coll[rev_idx] = cw_make(fwd_idx, prim);
        ws.fwd_coll[fwd_idx] = cw_make(rev_idx, prim);
      } else if (!rev_occ && fwd_occ) {
        const u32 barrier_pos = stalling_barrier_pos(*c.iv, fc);
        ws.rev_coll[rev_idx] = cw_make(fwd_idx, prim);
        if (cpos_rev < barrier_pos) ws.fwd_coll[fwd_idx] = cw_make(rev_idx, prim);
      }
    }
    carry_pos = wave::bcast(R, 63);
  }
  wave::sync_mem();
}

// correct_moves_for_lef_bar_collisions (reference: simulation_correct_moves.cpp:19-50)
MODLE_DEV void correct_moves_lef_bar(Cell& c) {
  Workspace& ws = c.ws;
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  for (u32 base = 0; base < n; base += 64) {
    const u32 i = base + lane;
    if (i < n) {
      const u32 rc = ws.rev_coll[i], fc = ws.fwd_coll[i];
      if (cw_occurred_as(rc, EV_LEF_BAR))
        ws.rev_moves[i] = (ws.rev_pos[i] - c.iv->bar_pos[cw_index(rc)]) - 1;
      if (cw_occurred_as(fc, EV_LEF_BAR))
        ws.fwd_moves[i] = (c.iv->bar_pos[cw_index(fc)] - ws.fwd_pos[i]) - 1;
    }
  }
  wave::sync_mem();
}

// correct_moves_for_primary_lef_lef_collisions (reference: simulation_correct_moves.cpp:53-121)
// Every unit takes part in at most one primary pair, so the two reference loops parallelise
// over LEF ids.
MODLE_DEV void correct_moves_primary(Cell& c) {
  Workspace& ws = c.ws;
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  for (u32 base = 0; base < n; base += 64) {
    const u32 r = base + lane;
    if (r < n) {
      const u32 rc = ws.rev_coll[r];
      if (cw_occurred_as(rc, EV_LEF_LEF_PRIMARY)) {
        const u32 f = cw_index(rc);
        const u32 fc = ws.fwd_coll[f];
        if (cw_occurred_as(fc, EV_LEF_LEF_PRIMARY)) {
          u32 p1, p2;
          lef_lef_collision_pos(ws.rev_pos[r], ws.fwd_pos[f], ws.rev_moves[r], ws.fwd_moves[f],
                                p1, p2);
          ws.rev_moves[r] = ws.rev_pos[r] - p1;
          ws.fwd_moves[f] = p2 - ws.fwd_pos[f];
        } else if (cw_occurred_as(fc, EV_LEF_BAR)) {
          ws.rev_moves[r] = ws.rev_pos[r] - (ws.fwd_pos[f] + ws.fwd_moves[f]) - 1;
        }
      }
    }
  }
  wave::sync_mem();
  for (u32 base = 0; base < n; base += 64) {
    const u32 f = base + lane;
    if (f < n) {
      const u32 fc = ws.fwd_coll[f];
      if (cw_occurred_as(fc, EV_LEF_LEF_PRIMARY)) {
        const u32 r = cw_index(fc);
        if (cw_occurred_as(ws.rev_coll[r], EV_LEF_BAR))
          ws.fwd_moves[f] = (ws.rev_pos[r] - ws.rev_moves[r]) - ws.fwd_pos[f] - 1;
      }
    }
  }
  wave::sync_mem();
}

// process_secondary_lef_lef_collisions (reference: simulation_detect_collisions.cpp:400-515).
// The pass is a chain: a stalled unit can stall its follower, which can stall the next one, and
// every candidate consumes one Bernoulli draw in rank order.  Each batch of 64 consecutive ranks
// is loaded into registers, a vector test discards the ranks that cannot be candidates and the
// rest are walked in order with lane broadcasts.  Ranks whose collision was avoided are
// appended to `list` (rank positions, visiting order) for fix_secondary.
template <bool FWD>
MODLE_DEV u32 process_secondary(Cell& c, BoundaryCounts bc, u32* list, u32 list_cap,
                                bool& overflow) {
  Workspace& ws = c.ws;
  const Params& p = *c.p;
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  const u32* rank = FWD ? ws.fwd_rank : ws.rev_rank;
  const u32* pos = FWD ? ws.fwd_pos : ws.rev_pos;
  u32* moves = FWD ? ws.fwd_moves : ws.rev_moves;
  u32* coll = FWD ? ws.fwd_coll : ws.rev_coll;
  const bool trials = p.p_bypass != 0.0;
  const f64 p_collide = 1.0 - p.p_bypass;
  u32 n_list = 0;
  // rev: i = max(1, n5) .. n-1 ascending, U1 = rank i-1 (blocker), U2 = rank i
  // fwd: i = (n - min(n3, n3-1) - 1) .. 1 descending, U2 = rank i (blocker), U1 = rank i-1
  // In both cases the "follower" is visited in order and its blocker is the previously visited
  // neighbour; `f` below indexes followers.
  const i64 f_first = FWD ? static_cast<i64>(bc.n3 == 0 ? n - 1 : n - bc.n3) - 1
                          : static_cast<i64>(umax(1u, bc.n5));
  const i64 f_last = FWD ? 0 : static_cast<i64>(n) - 1;  // inclusive
  const i64 count = FWD ? f_first - f_last + 1 : f_last - f_first + 1;
  if (count <= 0) return 0;
  // blocker state carried from one batch to the next
  u32 carry_pos, carry_move, carry_coll, carry_id;
  {
    const u32 kb = static_cast<u32>(FWD ? f_first + 1 : f_first - 1);
    carry_id = rank[kb];
    carry_pos = pos[carry_id];
    carry_move = moves[carry_id];
    carry_coll = coll[carry_id];
  }
  for (i64 done = 0; done < count; done += 64) {
    const i64 kk = FWD ? f_first - done - lane : f_first + done + lane;
    const bool act = done + lane < count;
    const u32 k = act ? static_cast<u32>(kk) : 0;
    const u32 id = act ? rank[k] : 0;
    const u32 P = act ? pos[id] : 0;
    u32 M = act ? moves[id] : 0;
    u32 C = act ? coll[id] : 0;
    const u32 M0 = M, C0 = C;
    // vector pre-filter: follower free and able to reach the blocker's current position
    const u32 bp_in = wave::shfl_up(P, 1);
    const u32 blocker_pos = lane > 0 ? bp_in : carry_pos;
    const bool pot = act && !cw_occurred(C) &&
                     (FWD ? static_cast<u64>(P) + M >= blocker_pos
                          : static_cast<u64>(P) - M <= blocker_pos);
    u64 todo = wave::ballot(pot);
    while (todo != 0) {
      const u32 l = static_cast<u32>(wave::ctz64(todo));
      todo &= todo - 1;
      // blocker = lane l-1 (current register state) or the carried unit
      const u32 bP = l > 0 ? wave::bcast(P, l - 1) : carry_pos;
      const u32 bM = l > 0 ? wave::bcast(M, l - 1) : carry_move;
      const u32 bC = l > 0 ? wave::bcast(C, l - 1) : carry_coll;
      const u32 bId = l > 0 ? wave::bcast(id, l - 1) : carry_id;
      const u32 fP = wave::bcast(P, l);
      const u32 fM = wave::bcast(M, l);
      const u32 fK = wave::bcast(k, l);
      if (!cw_occurred(bC)) continue;
      const bool geo = FWD ? (static_cast<u64>(fP) + fM >= static_cast<u64>(bP) + bM)
                           : (static_cast<u64>(fP) - fM <= static_cast<u64>(bP) - bM);
      if (!geo) continue;
      bool collide = true;
      if (trials) collide = bernoulli_raw(rng_next(c.g), p_collide);
      if (collide) {
        const u32 move = FWD ? (bP + bM) - fP : fP - (bP - bM);
        const u32 newM = umin(move, move - 1);
        if (lane == l) {
          M = newM;
          C = cw_make(bId, EV_COLLISION | EV_LEF_LEF_SECONDARY);
        }
      } else {
        if (lane == l) C = cw_make(bId, EV_LEF_LEF_SECONDARY);
        if (n_list < list_cap) {
          if (lane == 0) list[n_list] = fK;
        } else {
          overflow = true;
        }
        ++n_list;
      }
    }
    if (act && (M != M0 || C != C0)) {
      moves[id] = M;
      coll[id] = C;
    }
    carry_pos = wave::bcast(P, 63);
    carry_move = wave::bcast(M, 63);
    carry_coll = wave::bcast(C, 63);
    carry_id = wave::bcast(id, 63);
  }
  wave::sync_mem();
  return n_list;
}

// fix_secondary_lef_lef_collisions (reference: simulation_detect_collisions.cpp:517-644).
// Rare (one entry per avoided secondary collision); replayed sequentially, uniformly.
MODLE_DEV void fix_secondary_rev(Cell& c, const u32* list, u32 n_list) {
  Workspace& ws = c.ws;
  const u32 start = c.iv->start;
  const u32 sec = EV_LEF_LEF_SECONDARY;
  for (u32 q = 0; q < n_list; ++q) {  // list is in ascending rank order
    const u32 i = list[q];
    const u32 idx2 = ws.rev_rank[i];
    if (!cw_avoided_as(ws.rev_coll[idx2], sec)) continue;
    const u32 idx1 = ws.rev_rank[i - 1];
    const u32 pos1 = ws.rev_pos[idx1] - ws.rev_moves[idx1];
    u32 m2 = 0;
    if (ws.rev_pos[idx2] > pos1 + 1) m2 = ws.rev_pos[idx2] - (pos1 + 1);
    const u32 c2 = cw_make(idx1, EV_COLLISION | sec);
    const u32 p1 = ws.rev_pos[idx1], p2 = ws.rev_pos[idx2];
    const u32 np1 = umin(ws.fwd_pos[idx1], p2);
    const u32 np2 = umin(ws.fwd_pos[idx2], p1);
    const u32 c1 = ws.rev_coll[idx1];
    const u32 m1 = ws.rev_moves[idx1];
    wave::lockstep();
    // swapped collisions / moves, then re-clamp
    ws.rev_pos[idx1] = np1;
    ws.rev_pos[idx2] = np2;
    ws.rev_coll[idx1] = c2;
    ws.rev_coll[idx2] = c1;
    ws.rev_moves[idx1] = umin(np1 - start, m2);
    ws.rev_moves[idx2] = umin(np2 - start, m1);
    ws.rev_rank[i - 1] = idx2;
    ws.rev_rank[i] = idx1;
    wave::sync_mem();
  }
}

MODLE_DEV void fix_secondary_fwd(Cell& c, const u32* list, u32 n_list) {
  Workspace& ws = c.ws;
  const u32 last = c.iv->end - 1;
  const u32 sec = EV_LEF_LEF_SECONDARY;
  for (u32 q = n_list; q-- > 0;) {  // list is in descending rank order; the fix loop ascends
    const u32 i = list[q];
    const u32 idx1 = ws.fwd_rank[i];
    if (!cw_avoided_as(ws.fwd_coll[idx1], sec)) continue;
    const u32 idx2 = ws.fwd_rank[i + 1];
#ifdef MODLE_TRACE
    if (wave::lane() == 0 && getenv("MO_TRACE_FIX"))
      fprintf(stderr, "FIXF i=%u idx1=%u idx2=%u c1=%x c2=%x p1=%u p2=%u m1=%u m2=%u\n", i, idx1,
              idx2, ws.fwd_coll[idx1], ws.fwd_coll[idx2], ws.fwd_pos[idx1], ws.fwd_pos[idx2],
              ws.fwd_moves[idx1], ws.fwd_moves[idx2]);
#endif
    const u32 pos2 = ws.fwd_pos[idx2] + ws.fwd_moves[idx2];
    u32 m1 = 0;
    if (pos2 > ws.fwd_pos[idx1] + 1) m1 = pos2 - (ws.fwd_pos[idx1] + 1);
    const u32 c1 = cw_make(idx2, EV_COLLISION | sec);
    const u32 p1 = ws.fwd_pos[idx1], p2 = ws.fwd_pos[idx2];
    const u32 np1 = umax(ws.rev_pos[idx1], p2);
    const u32 np2 = umax(ws.rev_pos[idx2], p1);
    const u32 c2 = ws.fwd_coll[idx2];
    const u32 m2 = ws.fwd_moves[idx2];
    wave::lockstep();
    ws.fwd_pos[idx1] = np1;
    ws.fwd_pos[idx2] = np2;
    ws.fwd_coll[idx1] = c2;
    ws.fwd_coll[idx2] = c1;
    ws.fwd_moves[idx1] = umin(last - np1, m2);
    ws.fwd_moves[idx2] = umin(last - np2, m1);
    ws.fwd_rank[i] = idx2;
    ws.fwd_rank[i + 1] = idx1;
    wave::sync_mem();
  }
}

MODLE_DEV void build_sorted_positions(Cell& c, const u32* rank, const u32* pos, u32* out) {
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    if (k < n) out[k] = pos[rank[k]];
  }
  wave::sync_mem();
}

MODLE_DEV void clear_collisions(Cell& c) {
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  for (u32 base = 0; base < n; base += 64) {
    const u32 i = base + lane;
    if (i < n) {
      c.ws.rev_coll[i] = 0;
      c.ws.fwd_coll[i] = 0;
    }
  }
  wave::sync_mem();
}

// returns false when the per-wave list overflowed (the cell is then flagged as failed)
MODLE_DEV bool phase_process_collisions(Cell& c) {
  const BoundaryCounts bc = detect_boundaries(c);
  detect_lef_bar<false>(c, bc);
  detect_lef_bar<true>(c, bc);
  build_sorted_positions(c, c.ws.fwd_rank, c.ws.fwd_pos, c.ws.tmp_c);
  detect_primary(c, bc, c.ws.tmp_c);
  correct_moves_lef_bar(c);
  correct_moves_primary(c);
  bool overflow = false;
  u32* list_rev = c.lds.list;
  u32* list_fwd = c.lds.list + LIST_CAP / 2;
  const u32 nr = process_secondary<false>(c, bc, list_rev, LIST_CAP / 2, overflow);
  const u32 nf = process_secondary<true>(c, bc, list_fwd, LIST_CAP / 2, overflow);
  if (overflow) c.error = ERR_LIST_OVERFLOW;
  if (c.error != 0) return false;
  if (nr != 0) fix_secondary_rev(c, list_rev, nr);
  if (nf != 0) fix_secondary_fwd(c, list_fwd, nf);
  return true;
}

// =============================================================================================
// extrude + release_lefs (reference: simulation.cpp:498-521, 553-601)
// =============================================================================================
MODLE_DEV void phase_extrude_and_release(Cell& c, bool burnin_completed) {
  Workspace& ws = c.ws;
  const Params& p = *c.p;
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  const f64 base_p = burnin_completed ? p.p_release : p.p_release_burnin;
  for (u32 base = 0; base < n; base += 64) {
    const u32 i = base + lane;
    const bool act = i < n;
    const bool bnd = act && ws.epoch[i] != UNBOUND;
    f64 prob = 0.0;
    if (bnd) {
      const u32 rc = ws.rev_coll[i], fc = ws.fwd_coll[i];
      u32 hard = 0;
      if (cw_occurred_as(rc, EV_LEF_BAR)) hard += c.iv->bar_dir[cw_index(rc)] == DIR_REV;
      if (cw_occurred_as(fc, EV_LEF_BAR)) hard += c.iv->bar_dir[cw_index(fc)] == DIR_FWD;
      const f64 affinity =
          hard == 0 ? 1.0 : (hard == 1 ? 1.0 / p.soft_stall_mult : 1.0 / p.hard_stall_mult);
      prob = affinity * base_p;
    }
    const bool draws = bnd && prob != 0.0;
    const u64 dm = wave::ballot(draws);
    const u32 cnt = static_cast<u32>(wave::popc64(dm));
    rng_ensure(c.g, cnt);
    const u32 k = static_cast<u32>(wave::popc64(dm & lanemask_lt(lane)));
    const bool rel = draws && bernoulli_raw(rng_peek(c.g, c.g.pos + k), prob);
    c.g.pos += cnt;
    if (bnd) {
      if (rel) {
        ws.rev_pos[i] = UNBOUND;
        ws.fwd_pos[i] = UNBOUND;
        ws.epoch[i] = UNBOUND;
      } else {
        ws.rev_pos[i] -= ws.rev_moves[i];
        ws.fwd_pos[i] += ws.fwd_moves[i];
      }
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

// lef_within_bound / randomize_extrusion_unit_positions / pos_within_bound
// (reference: register_contacts.cpp:23-63)
MODLE_DEV bool sample_lef_pair(const Cell& c, u32 i, f64 u1, f64 u2, bool noisify, f64& p1,
                               f64& p2) {
  const Workspace& ws = c.ws;
  const Params& p = *c.p;
  const f64 n1 = noisify ? genextreme_from_canonical(u1, p.gev_mu, p.gev_sigma, p.gev_xi) : 0.0;
  const f64 a = static_cast<f64>(ws.rev_pos[i]) - n1;
  const f64 n2 = noisify ? genextreme_from_canonical(u2, p.gev_mu, p.gev_sigma, p.gev_xi) : 0.0;
  const f64 b = static_cast<f64>(ws.fwd_pos[i]) + n2;
  p1 = b < a ? b : a;
  p2 = b < a ? a : b;
  const f64 lo = static_cast<f64>(c.iv->start + 1), hi = static_cast<f64>(c.iv->end - 1);
  return p1 >= lo && p2 >= lo && p1 < hi && p2 < hi;
}

MODLE_DEV bool lef_samplable(const Cell& c, u32 i) {
  const Workspace& ws = c.ws;
  const u32 lo = c.iv->start + 1, hi = c.iv->end - 1;
  if (ws.epoch[i] == UNBOUND) return false;
  const u32 r = ws.rev_pos[i], f = ws.fwd_pos[i];
  return r > lo && r < hi && f > lo && f < hi;
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
  const u32 i = static_cast<u32>(r);
  if (!lef_samplable(c, i)) return e;  // consumed = 1
  const u32 nz = noisify ? 2u : 0u;
  const f64 u1 = noisify ? canonical_raw(rng_peek(c.g, q + 1)) : 0.0;
  const f64 u2 = noisify ? canonical_raw(rng_peek(c.g, q + 2)) : 0.0;
  f64 p1, p2;
  const bool inb = sample_lef_pair(c, i, u1, u2, noisify, p1, p2);
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
MODLE_DEV EventEval eval_event_exact(Cell& c, u64 lef_range, u64 lef_bucket, bool noisify) {
  EventEval e{0, false, false, 0, 0};
  const u64 r = lef_range == 0 ? 0 : uniform_int_exact(c.g, lef_range, lef_bucket);
  const u32 i = static_cast<u32>(r);
  if (!lef_samplable(c, i)) return e;
  const f64 u1 = noisify ? canonical_raw(rng_next(c.g)) : 0.0;
  const f64 u2 = noisify ? canonical_raw(rng_next(c.g)) : 0.0;
  f64 p1, p2;
  if (!sample_lef_pair(c, i, u1, u2, noisify, p1, p2)) return e;
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
MODLE_DEV u64 run_events(Cell& c, u64 n_events) {
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
    const u32 cntb = static_cast<u32>(umin64(64, remaining));
    rng_ensure(c.g, cntb * stride);
    const bool act = lane < cntb;
    EventEval e{stride, false, false, 0, 0};
    if (act) e = eval_event_fast<KIND>(c, c.g.pos + static_cast<u64>(lane) * stride, lef_range,
                                       lef_bucket, noisify);
    const u64 irregular = wave::ballot(act && (e.need_exact || e.consumed != stride));
    if (irregular == 0) {
      if (act && e.ok) commit_event<KIND>(c, e);
      registered += static_cast<u64>(wave::popc64(wave::ballot(act && e.ok)));
      c.g.pos += static_cast<u64>(cntb) * stride;
      remaining -= cntb;
    } else {
      const u32 f = static_cast<u32>(wave::ctz64(irregular));
      const bool commit = act && e.ok && lane < f;
      if (commit) commit_event<KIND>(c, e);
      registered += static_cast<u64>(wave::popc64(wave::ballot(commit)));
      c.g.pos += static_cast<u64>(f) * stride;
      const bool needx = wave::bcast(e.need_exact, f);
      if (!needx) {
        if (lane == f && e.ok) commit_event<KIND>(c, e);
        registered += wave::bcast(e.ok, f) ? 1 : 0;
        c.g.pos += wave::bcast(e.consumed, f);
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
MODLE_DEV void compute_loop_size_stats(Cell& c) {
  // reference: simulation.cpp:795-819 and stats/descriptive_impl.hpp:22-31, 63-101.  The mean is
  // a sum of integers below 2^53 (order independent); the squared deviations are accumulated
  // strictly left to right like std::accumulate.
  Workspace& ws = c.ws;
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  const u32 cap = c.p->hist_len;
  u64 part = 0;
  for (u32 base = 0; base < n; base += 64) {
    const u32 i = base + lane;
    if (i < n) part += static_cast<u64>(ws.fwd_pos[i] - ws.rev_pos[i]);
  }
#pragma unroll
  for (u32 s = 1; s < 64; s <<= 1) {
    const u64 o = wave::shfl_down(part, s);
    if (lane + s < 64) part += o;
  }
  const u64 total = wave::bcast(part, 0);
  const f64 avg = static_cast<f64>(total) / static_cast<f64>(n);
  f64* terms = reinterpret_cast<f64*>(ws.sort_keys);
  for (u32 base = 0; base < n; base += 64) {
    const u32 i = base + lane;
    if (i < n) {
      const f64 d = static_cast<f64>(static_cast<u64>(ws.fwd_pos[i] - ws.rev_pos[i])) - avg;
      terms[i] = d * d;
    }
  }
  wave::sync_mem();
  f64 ssd = 0.0;
  for (u32 i = 0; i < n; ++i) ssd = ssd + terms[i];
  const f64 std = wave::f_sqrt(ssd / static_cast<f64>(n));
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

MODLE_DEV bool evaluate_burnin(const Cell& c) {
  // reference: simulation.cpp:821-864
  const u32 cap = c.p->hist_len;
  if (c.hist_len != cap) return false;
  if (!series_is_stable(c, c.ws.hist)) return false;
  return series_is_stable(c, c.ws.hist + cap);
}

// =============================================================================================
// Cell driver (reference: simulation.cpp:896-986)
// =============================================================================================
MODLE_DEV void reset_cell_buffers(Cell& c) {
  // State::reset_buffers (reference: simulation.cpp:617-627)
  Workspace& ws = c.ws;
  const u32 L = c.n_lefs;
  const u32 lane = wave::lane();
  for (u32 base = 0; base < L; base += 64) {
    const u32 i = base + lane;
    if (i < L) {
      ws.rev_pos[i] = UNBOUND;
      ws.fwd_pos[i] = UNBOUND;
      ws.epoch[i] = UNBOUND;
      ws.rev_rank[i] = i;
      ws.fwd_rank[i] = i;
      ws.rev_moves[i] = 0;
      ws.fwd_moves[i] = 0;
      ws.rev_coll[i] = 0;
      ws.fwd_coll[i] = 0;
    }
  }
  wave::sync_mem();
}

// Simulates one (interval, cell) task on the calling wave.  Returns 0 or a non-zero status when
// an internal capacity was exceeded (the host turns that into an error).
MODLE_DEV u32 simulate_cell(const Params& p, const Interval& iv, const Task& task,
                            const Workspace& ws, const WaveLds& lds, CellResult& res,
                            u32 debug_stage = 0) {
#define MODLE_STAGE(k) \
  if (debug_stage == (k)) { res.epochs = (k); return 0; }
  Cell c;
  c.p = &p;
  c.iv = &iv;
  c.ws = ws;
  c.lds = lds;
  c.n_lefs = task.num_lefs;
  c.n_active = 0;
  c.hist_len = 0;
  c.hist_head = 0;
  c.error = 0;
  c.g.ring = lds.ring;
  c.g.jump = lds.jump_table;
  MODLE_STAGE(1)
  rng_init(c.g, task.prng);
  MODLE_STAGE(2)
  reset_cell_buffers(c);
  MODLE_STAGE(3)

  u64 epoch = 0, num_burnin_epochs = 0, num_contacts = 0;
  u64 sum_active = 0, events_done = 0, sim_epochs = 0;
  bool burnin_completed = false;
  u32 status = 0;
  const f64 lef_binding_rate_burnin =
      static_cast<f64>(task.num_lefs) / static_cast<f64>(p.burnin_target_epochs_for_lef_activation);

  barriers_init_states(c);
  MODLE_STAGE(4)
  if (p.skip_burnin) {
    c.n_active = c.n_lefs;
    burnin_completed = true;
  }
  bool first_ranking = true;
  for (;; ++epoch) {
    if (p.target_contact_density >= 0) {
      if (num_contacts >= task.num_target_contacts) break;
    } else if (epoch - num_burnin_epochs >= task.num_target_epochs) {
      break;
    }
    if (!burnin_completed) {
      // run_burnin (reference: simulation.cpp:866-894)
      do {
        ++num_burnin_epochs;
        if (c.n_active != c.n_lefs) {
          const u64 k = poisson_exact(c.g, lef_binding_rate_burnin);
          const u64 na = static_cast<u64>(c.n_active) + k;
          c.n_active = na < c.n_lefs ? static_cast<u32>(na) : c.n_lefs;
        } else {
          compute_loop_size_stats(c);
          burnin_completed = evaluate_burnin(c);
          burnin_completed = burnin_completed && epoch > p.min_burnin_epochs;
          if (!burnin_completed && epoch >= p.max_burnin_epochs) {
            burnin_completed = true;
            c.n_active = c.n_lefs;
          }
        }
      } while (c.n_active == 0);
    }

    MODLE_STAGE(5)
    if (debug_stage >= 100 && epoch == debug_stage - 100) {
      res.epochs = epoch;
      res.burnin_epochs = burnin_completed ? 1 : 0;
      res.num_contacts = c.n_active;
      return 0;
    }
    const u32 epoch32 = static_cast<u32>(epoch);
    phase_bind(c, epoch32);
    MODLE_STAGE(6)
    rank_update<false>(c, epoch32, first_ranking);
    rank_update<true>(c, epoch32, first_ranking);
    first_ranking = false;
    MODLE_STAGE(7)

    if (burnin_completed) {
      num_contacts += phase_sample_contacts(c, task.contacts_per_epoch, task.num_target_contacts,
                                            num_contacts, events_done);
      if (task.num_target_contacts != 0 && num_contacts >= task.num_target_contacts) break;
    }

    sum_active += c.n_active;
    ++sim_epochs;
    phase_generate_moves(c, burnin_completed);
    MODLE_STAGE(8)
    barriers_next_state(c);
    clear_collisions(c);
    MODLE_STAGE(9)
    if (!phase_process_collisions(c)) {
      status = c.error;
      break;
    }
#ifdef MODLE_TRACE
    if (wave::lane() == 0 && getenv("MO_TRACE") && getenv("MO_TRACE_EPOCH") &&
        (u64)atoll(getenv("MO_TRACE_EPOCH")) == epoch) {
      for (u32 i = 0; i < c.n_active; ++i)
        fprintf(stderr, "D %u %u %u %u %u %u:%u %u:%u\n", i, c.ws.rev_pos[i], c.ws.fwd_pos[i],
                c.ws.rev_moves[i], c.ws.fwd_moves[i], c.ws.rev_coll[i] >> 24,
                c.ws.rev_coll[i] & 0xFFFFFF, c.ws.fwd_coll[i] >> 24, c.ws.fwd_coll[i] & 0xFFFFFF);
    }
#endif
    MODLE_STAGE(10)
    phase_extrude_and_release(c, burnin_completed);
    MODLE_STAGE(11)
#ifdef MODLE_TRACE
    {
      u64 sr = 0, sf = 0;
      for (u32 i = 0; i < c.n_active; ++i) {
        sr += c.ws.rev_pos[i];
        sf += c.ws.fwd_pos[i];
      }
      if (wave::lane() == 0 && getenv("MO_TRACE"))
        fprintf(stderr, "T %llu %llu %llu %llu %u %llu\n", (unsigned long long)epoch,
                (unsigned long long)c.g.pos, (unsigned long long)sr, (unsigned long long)sf,
                c.n_active, (unsigned long long)num_contacts);
    }
#endif
  }

  res.epochs = epoch;
  res.burnin_epochs = num_burnin_epochs;
  res.num_contacts = num_contacts;
  res.raws_consumed = c.g.pos;
  res.prng_final[0] = res.prng_final[1] = res.prng_final[2] = res.prng_final[3] = 0;
  res.sum_active_lefs = sum_active;
  res.sampling_events = events_done;
  res.sim_epochs = sim_epochs;
  return status;
}

// =============================================================================================
// Phase-level entry point (mirrors Simulation::test_* hooks, reference: simulation.hpp:413-567)
// =============================================================================================
constexpr u32 PH_RANK = 0x001, PH_RANK_INIT = 0x002, PH_ADJUST = 0x004, PH_CLAMP = 0x008,
              PH_BOUNDARIES = 0x010, PH_LEF_BAR = 0x020, PH_PRIMARY = 0x040,
              PH_CORRECT_LEF_BAR = 0x080, PH_CORRECT_PRIMARY = 0x100, PH_SECONDARY = 0x200,
              PH_FIX_SECONDARY = 0x400, PH_USE_BOUNDARY_COUNTS = 0x800;

// rank positions whose unit carries an "avoided secondary collision" mark, in the order
// process_secondary would have produced them
template <bool FWD>
MODLE_DEV u32 collect_avoided(Cell& c, u32* list, u32 cap) {
  const u32 n = c.n_active;
  const u32 lane = wave::lane();
  const u32* rank = FWD ? c.ws.fwd_rank : c.ws.rev_rank;
  const u32* coll = FWD ? c.ws.fwd_coll : c.ws.rev_coll;
  u32 cnt = 0;
  for (u32 base = 0; base < n; base += 64) {
    const u32 off = base + lane;
    const bool act = off < n;
    const u32 k = FWD ? (n - 1 - off) : off;
    const bool hit = act && cw_avoided_as(coll[rank[act ? k : 0]], EV_LEF_LEF_SECONDARY) &&
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
                              const WaveLds& lds, u32 mask, u32 n, const u64 prng[4],
                              u64& raws_consumed) {
  Cell c;
  c.p = &p;
  c.iv = &iv;
  c.ws = ws;
  c.lds = lds;
  c.n_lefs = n;
  c.n_active = n;
  c.hist_len = 0;
  c.hist_head = 0;
  c.error = 0;
  c.g.ring = lds.ring;
  c.g.jump = lds.jump_table;
  rng_init(c.g, prng);
  const u32 lane = wave::lane();
  if (mask & PH_RANK) {
    if (mask & PH_RANK_INIT) {
      for (u32 base = 0; base < n; base += 64) {
        const u32 i = base + lane;
        if (i < n) {
          c.ws.rev_rank[i] = i;
          c.ws.fwd_rank[i] = i;
        }
      }
      wave::sync_mem();
    }
    rank_update<false>(c, 0, true);
    rank_update<true>(c, 0, true);
  }
  if (mask & PH_ADJUST) {
    adjust_moves_rev(c, c.ws.rev_moves, c.ws.tmp_a);
    adjust_moves_fwd(c, c.ws.fwd_moves, c.ws.tmp_b);
    if (mask & PH_CLAMP) {
      clamp_moves(c, c.ws.tmp_a, c.ws.tmp_b);
    } else {
      for (u32 base = 0; base < n; base += 64) {
        const u32 i = base + lane;
        if (i < n) {
          c.ws.rev_moves[i] = c.ws.tmp_a[i];
          c.ws.fwd_moves[i] = c.ws.tmp_b[i];
        }
      }
      wave::sync_mem();
    }
  } else if (mask & PH_CLAMP) {
    for (u32 base = 0; base < n; base += 64) {
      const u32 i = base + lane;
      if (i < n) {
        c.ws.tmp_a[i] = c.ws.rev_moves[i];
        c.ws.tmp_b[i] = c.ws.fwd_moves[i];
      }
    }
    wave::sync_mem();
    clamp_moves(c, c.ws.tmp_a, c.ws.tmp_b);
  }
  BoundaryCounts bc{0, 0};
  if (mask & PH_BOUNDARIES) {
    const BoundaryCounts got = detect_boundaries(c);
    if (mask & PH_USE_BOUNDARY_COUNTS) bc = got;
  }
  if (mask & PH_LEF_BAR) {
    detect_lef_bar<false>(c, bc);
    detect_lef_bar<true>(c, bc);
  }
  if (mask & PH_PRIMARY) {
    build_sorted_positions(c, c.ws.fwd_rank, c.ws.fwd_pos, c.ws.tmp_c);
    detect_primary(c, bc, c.ws.tmp_c);
  }
  if (mask & PH_CORRECT_LEF_BAR) correct_moves_lef_bar(c);
  if (mask & PH_CORRECT_PRIMARY) correct_moves_primary(c);
  u32* list_rev = c.lds.list;
  u32* list_fwd = c.lds.list + LIST_CAP / 2;
  u32 nr = 0, nf = 0;
  bool overflow = false;
  if (mask & PH_SECONDARY) {
    nr = process_secondary<false>(c, bc, list_rev, LIST_CAP / 2, overflow);
    nf = process_secondary<true>(c, bc, list_fwd, LIST_CAP / 2, overflow);
  }
  if (mask & PH_FIX_SECONDARY) {
    if (!(mask & PH_SECONDARY)) {
      nr = collect_avoided<false>(c, list_rev, LIST_CAP / 2);
      nf = collect_avoided<true>(c, list_fwd, LIST_CAP / 2);
    }
    if (nr != 0) fix_secondary_rev(c, list_rev, nr);
    if (nf != 0) fix_secondary_fwd(c, list_fwd, nf);
  }
  raws_consumed = c.g.pos;
  return overflow ? ERR_LIST_OVERFLOW : c.error;
}

}  // namespace modle_dev

/* modle_oracle.c -- CPU restatement of MoDLE's per-cell loop-extrusion epoch loop.
 *
 * TEST INFRASTRUCTURE ONLY (see modle_oracle.h).  Plain C11, glibc libm, no other dependency.
 * Compile with -ffp-contract=off (the reference's x86-64 release builds carry no FMA).
 *
 * Citations are file:line relative to /root/reference.
 */
#include "modle_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zig_tables.h"
/* log / exp / pow: the reference calls glibc here.  The oracle and the device code compile the
 * SAME software routines (modle_amd/csrc/modle_math.h) so that every decision that
 * depends on them is identical on both sides by construction; sqrt is IEEE-exact everywhere. */
#include "../modle_amd/csrc/modle_math.h"
/* -DMO_USE_LIBM (make libmodle_oracle_libm.so): the same restatement calling the C library's
 * log / exp / pow like the reference does (glibc).  Never the parity oracle -- the device has no
 * glibc -- but the instrument that MEASURES how often a rejection test or a noise draw decides
 * differently between the shared routines and glibc (tools/libm_flip_rate.py). */
#ifdef MO_USE_LIBM
#include <math.h>
#define mm_log(x) log(x)
#define mm_exp(x) exp(x)
#define mm_pow(x, y) pow((x), (y))
#endif

#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))

/* ============================================================================================
 * PRNG: xoshiro256++ seeded through SplitMix64 (xoshiro-cpp 1.1, Vigna's public-domain
 * algorithms; reference use: src/common/include/modle/common/random.hpp:26-32)
 * ========================================================================================== */
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void mo_prng_seed(mo_prng_t* g, uint64_t seed) {
  /* random.hpp:28-29: SplitMix64(seed).generateSeedSequence<4>() */
  for (int i = 0; i < 4; ++i) {
    uint64_t z = (seed += UINT64_C(0x9e3779b97f4a7c15));
    z = (z ^ (z >> 30)) * UINT64_C(0xbf58476d1ce4e5b9);
    z = (z ^ (z >> 27)) * UINT64_C(0x94d049bb133111eb);
    g->s[i] = z ^ (z >> 31);
  }
  g->count = 0;
}

/* Generator policy of the in-cell stream (test infrastructure switch, process wide):
 * 0 = the reference's xoshiro256++ stream (every parity claim), 1 = the counter-based PHILOX
 * policy of the device code (modle_amd/csrc/sim_rng.h, MODLE_RNG_PHILOX): output p of a cell's
 * stream is one half of Philox4x32-10(counter = (p >> 1, s1 ^ s3), key = s0 ^ s2).  Seeding and
 * the per-cell jump() are the same in both policies. */
static uint64_t xoshiro_step_impl(uint64_t* s);
static inline uint64_t xoshiro_step(uint64_t* s) { return xoshiro_step_impl(s); }
static int g_rng_policy = 0;
void mo_set_rng_policy(int philox) { g_rng_policy = philox; }
int mo_get_rng_policy(void) { return g_rng_policy; }

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

/* the round function on its own, for the Random123 known-answer vectors */
void mo_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c[4] = {counter[0], counter[1], counter[2], counter[3]};
  philox4x32_10(c, key[0], key[1]);
  for (int i = 0; i < 4; ++i) out[i] = c[i];
}

uint64_t mo_prng_next(mo_prng_t* g) {
  uint64_t* s = g->s;
  if (g_rng_policy == 1) {
    const uint64_t p = g->count++;
    const uint64_t q = p >> 1, key = s[0] ^ s[2], hi = s[1] ^ s[3];
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
    philox4x32_10(c, (uint32_t)key, (uint32_t)(key >> 32));
    return (p & 1) ? (((uint64_t)c[3] << 32) | c[2]) : (((uint64_t)c[1] << 32) | c[0]);
  }
  ++g->count;
  return xoshiro_step(s);
}

static uint64_t xoshiro_step_impl(uint64_t* s) {
  const uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
  const uint64_t t = s[1] << 17;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl64(s[3], 45);
  return result;
}

void mo_prng_jump(mo_prng_t* g) {
  /* 2^128 steps; used once per cell by the scheduler (scheduler_simulate.cpp:158) */
  static const uint64_t JUMP[4] = {UINT64_C(0x180ec6d33cfd0aba), UINT64_C(0xd5a61266f0c9392c),
                                   UINT64_C(0xa9582618e03fc9aa), UINT64_C(0x39abdc4529b1661c)};
  uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  const uint64_t count = g->count;
  for (int i = 0; i < 4; ++i) {
    for (int b = 0; b < 64; ++b) {
      if (JUMP[i] & (UINT64_C(1) << b)) {
        s0 ^= g->s[0];
        s1 ^= g->s[1];
        s2 ^= g->s[2];
        s3 ^= g->s[3];
      }
      (void)xoshiro_step(g->s); /* the jump is a property of the seed sequence: always xoshiro */
    }
  }
  g->s[0] = s0;
  g->s[1] = s1;
  g->s[2] = s2;
  g->s[3] = s3;
  g->count = count;
}

/* ============================================================================================
 * Distributions -- Boost.Random 1.88 semantics on a full-range 64-bit engine (aliases:
 * random.hpp:34-53).  Restated from the published algorithms; Boost itself is not available
 * here, so these draws are "parity unpinned" against an official binary (modle_oracle.h).
 * ========================================================================================== */
static const double TWO64 = 18446744073709551616.0;      /* 2^64 */
static const double TWO_M64 = 5.42101086242752217e-20;   /* 2^-64 */
static const double TWO_M56 = 1.387778780781445675529539585113525390625e-17; /* 2^-56 */

/* boost::random::bernoulli_distribution<double>: p == 0 draws nothing;
 * otherwise double(raw) <= p * double(max - min) with max - min = 2^64 - 1 -> 2^64 */
int mo_bernoulli(mo_prng_t* g, double p) {
  if (p == 0.0) return 0;
  return (double)mo_prng_next(g) <= p * TWO64;
}

/* boost::random::generate_canonical<double, 53>: one 64-bit draw, divided by 2^64; a result
 * of exactly 1 is nudged down by epsilon/2 */
double mo_canonical(mo_prng_t* g) {
  double r = (double)mo_prng_next(g) / TWO64;
  if (r == 1.0) r -= 2.220446049250313e-16 / 2;
  return r;
}

/* boost::random::uniform_01<double>: raw * 2^-64, redrawn while the product rounds to 1 */
double mo_uniform_01(mo_prng_t* g) {
  for (;;) {
    const double r = (double)mo_prng_next(g) * TWO_M64;
    if (r < 1.0) return r;
  }
}

/* boost::random::uniform_int_distribution<uint64_t>{lo, hi}: bucket rejection */
uint64_t mo_uniform_int(mo_prng_t* g, uint64_t lo, uint64_t hi) {
  const uint64_t range = hi - lo;
  if (range == 0) return lo;
  if (range == UINT64_MAX) return mo_prng_next(g) + lo;
  uint64_t bucket = UINT64_MAX / (range + 1);
  if (UINT64_MAX % (range + 1) == range) ++bucket;
  for (;;) {
    const uint64_t r = mo_prng_next(g) / bucket;
    if (r <= range) return r + lo;
  }
}

/* generate_int_float_pair<double, 8> on a 64-bit engine: low 8 bits -> bucket, the 53 bits
 * above bit 10 -> abscissa in [0, 1) */
static inline double int_float_pair8(mo_prng_t* g, int* bucket) {
  uint64_t u = mo_prng_next(g);
  *bucket = (int)(u & 0xFF);
  u &= ~((UINT64_C(1) << 11) - 1);
  return (double)(u >> 8) * TWO_M56;
}

/* boost::random::detail::unit_exponential_distribution (256-layer ziggurat) */
static double unit_exponential(mo_prng_t* g) {
  double shift = 0.0;
  for (;;) {
    int i;
    const double u = int_float_pair8(g, &i);
    const double x = u * ZIG_EXP_X[i];
    if (x < ZIG_EXP_X[i + 1]) return shift + x;
    if (i == 0) {
      shift += ZIG_EXP_X[1];
    } else {
      const double y01 = mo_uniform_01(g);
      const double y = ZIG_EXP_Y[i] + y01 * (ZIG_EXP_Y[i + 1] - ZIG_EXP_Y[i]);
      const double y_above_ubound = (ZIG_EXP_X[i] - ZIG_EXP_X[i + 1]) * y01 - (ZIG_EXP_X[i] - x);
      const double y_above_lbound =
          y - (ZIG_EXP_Y[i + 1] + (ZIG_EXP_X[i + 1] - x) * ZIG_EXP_Y[i + 1]);
      if (y_above_ubound < 0 && (y_above_lbound < 0 || y < mm_exp(-x))) return x + shift;
    }
  }
}

/* boost::random::detail::unit_normal_distribution (128-layer ziggurat, exponential tail) */
static double unit_normal(mo_prng_t* g) {
  for (;;) {
    int b;
    const double u = int_float_pair8(g, &b);
    const int sign = (b & 1) * 2 - 1;
    const int i = b >> 1;
    const double x = u * ZIG_NORM_X[i];
    if (x < ZIG_NORM_X[i + 1]) return x * sign;
    if (i == 0) {
      const double tail_start = ZIG_NORM_X[1];
      for (;;) {
        const double tx = unit_exponential(g) / tail_start;
        const double ty = unit_exponential(g);
        if (2 * ty > tx * tx) return (tx + tail_start) * sign;
      }
    }
    const double y01 = mo_uniform_01(g);
    const double y = ZIG_NORM_Y[i] + y01 * (ZIG_NORM_Y[i + 1] - ZIG_NORM_Y[i]);
    const double chord = (ZIG_NORM_X[i] - ZIG_NORM_X[i + 1]) * y01 - (ZIG_NORM_X[i] - x);
    const double tangent = y - (ZIG_NORM_Y[i] + (ZIG_NORM_X[i] - x) * ZIG_NORM_Y[i] * ZIG_NORM_X[i]);
    double y_above_ubound, y_above_lbound;
    if (ZIG_NORM_X[i] >= 1) { /* convex side */
      y_above_ubound = chord;
      y_above_lbound = tangent;
    } else { /* concave side */
      y_above_lbound = chord;
      y_above_ubound = tangent;
    }
    if (y_above_ubound < 0 && (y_above_lbound < 0 || y < mm_exp(-(x * x / 2)))) return x * sign;
  }
}

/* boost::random::normal_distribution<double>{mean, sigma}: a fresh object per draw
 * (simulation.cpp:291) => no cached variate */
double mo_normal(mo_prng_t* g, double mean, double sigma) { return unit_normal(g) * sigma + mean; }

/* boost::random::poisson_distribution<size_t, double>: inversion for mean < 10, else PTRD
 * (Hoermann 1993) */
uint64_t mo_poisson(mo_prng_t* g, double mean) {
  static const double log_fact[10] = {0.0,
                                      0.0,
                                      0.69314718055994529,
                                      1.7917594692280550,
                                      3.1780538303479458,
                                      4.7874917427820458,
                                      6.5792512120101012,
                                      8.5251613610654147,
                                      10.604602902745251,
                                      12.801827480081469};
  if (mean < 10) {
    double p = mm_exp(-mean);
    uint64_t x = 0;
    double u = mo_uniform_01(g);
    while (u > p) {
      u = u - p;
      ++x;
      p = mean * p / (double)x;
    }
    return x;
  }
  const double smu = sqrt(mean);
  const double b = 0.931 + 2.53 * smu;
  const double a = -0.059 + 0.02483 * b;
  const double inv_alpha = 1.1239 + 1.1328 / (b - 3.4);
  const double v_r = 0.9277 - 3.6224 / (b - 2);
  for (;;) {
    double u;
    double v = mo_uniform_01(g);
    if (v <= 0.86 * v_r) {
      u = v / v_r - 0.43;
      return (uint64_t)floor((2 * a / (0.5 - fabs(u)) + b) * u + mean + 0.445);
    }
    if (v >= v_r) {
      u = mo_uniform_01(g) - 0.5;
    } else {
      u = v / v_r - 0.93;
      u = ((u < 0) ? -0.5 : 0.5) - u;
      v = mo_uniform_01(g) * v_r;
    }
    const double us = 0.5 - fabs(u);
    if (us < 0.013 && v > us) continue;
    const double k = floor((2 * a / us + b) * u + mean + 0.445);
    v = v * inv_alpha / (a / (us * us) + b);
    const double log_sqrt_2pi = 0.91893853320467267;
    if (k >= 10) {
      if (mm_log(v * smu) <= (k + 0.5) * mm_log(mean / k) - mean - log_sqrt_2pi + k -
                              (1 / 12. - (1 / 360. - 1 / (1260. * k * k)) / (k * k)) / k) {
        return (uint64_t)k;
      }
    } else if (k >= 0) {
      if (mm_log(v) <= k * mm_log(mean) - mean - log_fact[(int)k]) return (uint64_t)k;
    }
  }
}

static double binom_fc(int64_t k) {
  static const double table[10] = {0.08106146679532726, 0.04134069595540929, 0.02767792568499834,
                                   0.02079067210376509, 0.01664469118982119, 0.01387612882307075,
                                   0.01189670994589177, 0.01041126526197209, 0.009255462182712733,
                                   0.008330563433362871};
  if (k < 10) return table[k];
  const double ikp1 = 1.0 / (double)(k + 1);
  return (1.0 / 12 - (1.0 / 360 - (1.0 / 1260) * (ikp1 * ikp1)) * (ikp1 * ikp1)) * ikp1;
}

static int64_t binom_invert(mo_prng_t* g, int64_t t, double p, double q_n) {
  const double q = 1 - p;
  const double s = p / q;
  const double a = (double)(t + 1) * s;
  double r = q_n;
  double u = mo_uniform_01(g);
  int64_t x = 0;
  while (u > r) {
    u = u - r;
    ++x;
    const double r1 = ((a / (double)x) - s) * r;
    if (r1 < 2.220446049250313e-16 && r1 < r) break;
    r = r1;
  }
  return x;
}

/* boost::random::binomial_distribution<ptrdiff_t, double>{t, p}: inversion when
 * (t+1)*min(p,1-p) < 11, else BTRD (Hoermann 1993) */
int64_t mo_binomial(mo_prng_t* g, int64_t t, double p_) {
  const double p = (0.5 < p_) ? (1 - p_) : p_;
  const int64_t m = (int64_t)((double)(t + 1) * p);
  if (m < 11) {
    const double q_n = mm_pow(1 - p, (double)t);
    const int64_t x = binom_invert(g, t, p, q_n);
    return (0.5 < p_) ? t - x : x;
  }
  const double r = p / (1 - p);
  const double nr = (double)(t + 1) * r;
  const double npq = (double)t * p * (1 - p);
  const double sqrt_npq = sqrt(npq);
  const double b = 1.15 + 2.53 * sqrt_npq;
  const double a = -0.0873 + 0.0248 * b + 0.01 * p;
  const double c = (double)t * p + 0.5;
  const double alpha = (2.83 + 5.1 / b) * sqrt_npq;
  const double v_r = 0.92 - 4.2 / b;
  const double u_rv_r = 0.86 * v_r;
  int64_t k;
  for (;;) {
    double u;
    double v = mo_uniform_01(g);
    if (v <= u_rv_r) {
      u = v / v_r - 0.43;
      k = (int64_t)floor((2 * a / (0.5 - fabs(u)) + b) * u + c);
      break;
    }
    if (v >= v_r) {
      u = mo_uniform_01(g) - 0.5;
    } else {
      u = v / v_r - 0.93;
      u = ((u < 0) ? -0.5 : 0.5) - u;
      v = mo_uniform_01(g) * v_r;
    }
    const double us = 0.5 - fabs(u);
    k = (int64_t)floor((2 * a / us + b) * u + c);
    if (k < 0 || k > t) continue;
    v = v * alpha / (a / (us * us) + b);
    const double km = (double)llabs(k - m);
    if (km <= 15) {
      double f = 1;
      if (m < k) {
        int64_t i = m;
        do {
          ++i;
          f = f * (nr / (double)i - r);
        } while (i != k);
      } else if (m > k) {
        int64_t i = k;
        do {
          ++i;
          v = v * (nr / (double)i - r);
        } while (i != m);
      }
      if (v <= f) break;
      continue;
    }
    v = mm_log(v);
    const double rho = (km / npq) * (((km / 3. + 0.625) * km + 1. / 6) / npq + 0.5);
    const double tt = -km * km / (2 * npq);
    if (v < tt - rho) break;
    if (v > tt + rho) continue;
    const int64_t nm = t - m + 1;
    const double h =
        ((double)m + 0.5) * mm_log((double)(m + 1) / (r * (double)nm)) + binom_fc(m) + binom_fc(t - m);
    const int64_t nk = t - k + 1;
    if (v <= h + (double)(t + 1) * mm_log((double)nm / (double)nk) +
                 ((double)k + 0.5) * mm_log((double)nk * r / (double)(k + 1)) - binom_fc(k) -
                 binom_fc(t - k)) {
      break;
    }
  }
  return (0.5 < p_) ? t - k : k;
}

/* genextreme_value_distribution<double> (genextreme_value_distribution.hpp:87-105) */
double mo_genextreme(mo_prng_t* g, double mu, double sigma, double xi) {
  if (xi == 0.0) return (mu - sigma) * mm_log(-mm_log(mo_canonical(g)));
  return mu + (sigma * (1.0 - mm_pow(-mm_log(mo_canonical(g)), xi))) / xi;
}

/* ============================================================================================
 * XXH3-64 (xxHash 0.8.x, inputs of at most 240 bytes) -- GenomicInterval::hash
 * (src/libmodle/internal/genome.cpp:201-224) streams name || size || start || end into an
 * XXH3 state reset with the user seed; for <= 240 bytes the digest equals the one-shot hash.
 * ========================================================================================== */
static const uint8_t XXH3_SECRET[192] = {
    0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c,
    0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f,
    0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21,
    0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6, 0x81, 0x3a, 0x26, 0x4c,
    0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65, 0x8b, 0x1b, 0x53, 0x2e, 0xa3,
    0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19, 0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8,
    0xa8, 0xfa, 0x76, 0x3f, 0xe3, 0x9c, 0x34, 0x3f, 0xf9, 0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d,
    0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31, 0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64,
    0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb, 0xc3, 0xc5, 0x81, 0xa0, 0xff, 0xfa, 0x13, 0x63, 0xeb,
    0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0, 0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e,
    0x2b, 0x16, 0xbe, 0x58, 0x7d, 0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce,
    0x45, 0xcb, 0x3a, 0x8f, 0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e,
};
#define P32_1 UINT64_C(0x9E3779B1)
#define P32_2 UINT64_C(0x85EBCA77)
#define P32_3 UINT64_C(0xC2B2AE3D)
#define P64_1 UINT64_C(0x9E3779B185EBCA87)
#define P64_2 UINT64_C(0xC2B2AE3D27D4EB4F)
#define P64_3 UINT64_C(0x165667B19E3779F9)
#define P64_4 UINT64_C(0x85EBCA77C2B2AE63)
#define P64_5 UINT64_C(0x27D4EB2F165667C5)
#define PMX1 UINT64_C(0x165667919E3779F9)
#define PMX2 UINT64_C(0x9FB21C651E98DF25)

static inline uint64_t rd64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
static inline uint32_t rd32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
static inline uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }
static inline uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
static inline uint64_t mul128_fold64(uint64_t a, uint64_t b) {
  const __uint128_t m = (__uint128_t)a * b;
  return (uint64_t)m ^ (uint64_t)(m >> 64);
}
static inline uint64_t xxh64_avalanche(uint64_t h) {
  h ^= h >> 33;
  h *= P64_2;
  h ^= h >> 29;
  h *= P64_3;
  h ^= h >> 32;
  return h;
}
static inline uint64_t xxh3_avalanche(uint64_t h) {
  h ^= h >> 37;
  h *= PMX1;
  h ^= h >> 32;
  return h;
}
static inline uint64_t xxh3_rrmxmx(uint64_t h, uint64_t len) {
  h ^= rotl64(h, 49) ^ rotl64(h, 24);
  h *= PMX2;
  h ^= (h >> 35) + len;
  h *= PMX2;
  return h ^ (h >> 28);
}
static inline uint64_t xxh3_mix16(const uint8_t* in, const uint8_t* sec, uint64_t seed) {
  return mul128_fold64(rd64(in) ^ (rd64(sec) + seed), rd64(in + 8) ^ (rd64(sec + 8) - seed));
}

uint64_t mo_xxh3_64(const void* data, size_t len, uint64_t seed) {
  const uint8_t* in = (const uint8_t*)data;
  const uint8_t* sec = XXH3_SECRET;
  if (len <= 16) {
    if (len > 8) {
      const uint64_t bitflip1 = (rd64(sec + 24) ^ rd64(sec + 32)) + seed;
      const uint64_t bitflip2 = (rd64(sec + 40) ^ rd64(sec + 48)) - seed;
      const uint64_t lo = rd64(in) ^ bitflip1;
      const uint64_t hi = rd64(in + len - 8) ^ bitflip2;
      const uint64_t acc = len + bswap64(lo) + hi + mul128_fold64(lo, hi);
      return xxh3_avalanche(acc);
    }
    if (len >= 4) {
      seed ^= (uint64_t)bswap32((uint32_t)seed) << 32;
      const uint32_t in1 = rd32(in);
      const uint32_t in2 = rd32(in + len - 4);
      const uint64_t bitflip = (rd64(sec + 8) ^ rd64(sec + 16)) - seed;
      const uint64_t in64 = in2 + (((uint64_t)in1) << 32);
      return xxh3_rrmxmx(in64 ^ bitflip, len);
    }
    if (len > 0) {
      const uint8_t c1 = in[0], c2 = in[len >> 1], c3 = in[len - 1];
      const uint32_t combined =
          ((uint32_t)c1 << 16) | ((uint32_t)c2 << 24) | ((uint32_t)c3 << 0) | ((uint32_t)len << 8);
      const uint64_t bitflip = (rd32(sec) ^ rd32(sec + 4)) + seed;
      return xxh64_avalanche(combined ^ bitflip);
    }
    return xxh64_avalanche(seed ^ (rd64(sec + 56) ^ rd64(sec + 64)));
  }
  if (len <= 128) {
    uint64_t acc = len * P64_1;
    if (len > 32) {
      if (len > 64) {
        if (len > 96) {
          acc += xxh3_mix16(in + 48, sec + 96, seed);
          acc += xxh3_mix16(in + len - 64, sec + 112, seed);
        }
        acc += xxh3_mix16(in + 32, sec + 64, seed);
        acc += xxh3_mix16(in + len - 48, sec + 80, seed);
      }
      acc += xxh3_mix16(in + 16, sec + 32, seed);
      acc += xxh3_mix16(in + len - 32, sec + 48, seed);
    }
    acc += xxh3_mix16(in + 0, sec + 0, seed);
    acc += xxh3_mix16(in + len - 16, sec + 16, seed);
    return xxh3_avalanche(acc);
  }
  if (len <= 240) {
    uint64_t acc = len * P64_1;
    const size_t nb_rounds = len / 16;
    for (size_t i = 0; i < 8; ++i) acc += xxh3_mix16(in + 16 * i, sec + 16 * i, seed);
    acc = xxh3_avalanche(acc);
    for (size_t i = 8; i < nb_rounds; ++i)
      acc += xxh3_mix16(in + 16 * i, sec + 16 * (i - 8) + 3, seed);
    acc += xxh3_mix16(in + len - 16, sec + 136 - 17, seed);
    return xxh3_avalanche(acc);
  }
  return 0; /* longer inputs are not needed by the path */
}

uint64_t mo_interval_hash(const char* chrom_name, uint64_t chrom_size, uint64_t start,
                          uint64_t end, uint64_t seed) {
  /* genome.cpp:208-215: name bytes, then size/start/end as native (little-endian) u64 */
  uint8_t buf[240];
  size_t n = strlen(chrom_name);
  if (n > 240 - 24) n = 240 - 24;
  memcpy(buf, chrom_name, n);
  memcpy(buf + n, &chrom_size, 8);
  memcpy(buf + n + 8, &start, 8);
  memcpy(buf + n + 16, &end, 8);
  return mo_xxh3_64(buf, n + 24, seed);
}

/* ============================================================================================
 * Derived quantities (simulation.cpp:1076-1090, contact_matrix_dense_impl.hpp:40-44)
 * ========================================================================================== */
uint64_t mo_compute_num_lefs(const mo_params_t* p, uint64_t size_bp) {
  const double size_mbp = (double)size_bp / 1.0e6;
  const uint64_t n = (uint64_t)round(p->number_of_lefs_per_mbp * size_mbp);
  return MAX((uint64_t)1, n);
}

uint64_t mo_compute_contacts_per_epoch(const mo_params_t* p, uint64_t nlefs) {
  const double speed = (double)(p->rev_extrusion_speed + p->fwd_extrusion_speed);
  const double prob = speed / (double)p->contact_sampling_interval;
  return (uint64_t)fmax(1.0, round((double)nlefs * prob));
}

void mo_matrix_shape(const mo_params_t* p, uint64_t size_bp, uint64_t* nrows, uint64_t* ncols) {
  const uint64_t nr = (p->diagonal_width + p->bin_size - 1) / p->bin_size;
  const uint64_t nc = (size_bp + p->bin_size - 1) / p->bin_size;
  *nrows = MIN(nr, nc);
  *ncols = nc;
}

void mo_make_tasks(const mo_params_t* p, const char* chrom_name, uint64_t chrom_size,
                   uint64_t start, uint64_t end, uint64_t first_task_id, mo_task_t* tasks) {
  /* scheduler_simulate.cpp:108, 127-159 */
  mo_prng_t g;
  mo_prng_seed(&g, mo_interval_hash(chrom_name, chrom_size, start, end, p->seed));
  const uint64_t nlefs = mo_compute_num_lefs(p, end - start);
  uint64_t nrows, ncols;
  mo_matrix_shape(p, end - start, &nrows, &ncols);
  const uint64_t npixels = nrows * ncols;
  const uint64_t tot = (uint64_t)round((double)npixels * p->target_contact_density);
  const uint64_t per_cell = (tot + p->num_cells - 1) / p->num_cells;
  uint64_t rolling = 0;
  for (uint64_t c = 0; c < p->num_cells; ++c) {
    const uint64_t n = MIN(per_cell, tot - rolling);
    rolling += n;
    tasks[c].id = first_task_id + c;
    tasks[c].cell_id = c;
    tasks[c].num_target_epochs = p->target_simulation_epochs;
    tasks[c].num_target_contacts = n;
    tasks[c].num_lefs = nlefs;
    memcpy(tasks[c].prng, g.s, sizeof(g.s));
    mo_prng_jump(&g);
  }
}

/* extrusion_barriers_impl.hpp:106-128 */
static double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }
double mo_stp_active_from_occupancy(double stp_inactive, double occupancy) {
  if (occupancy == 0) return 0.0;
  const double tp_i2a = 1.0 - stp_inactive;
  const double tp_a2i = (tp_i2a - (occupancy * tp_i2a)) / occupancy;
  return clamp01(1.0 - tp_a2i);
}
double mo_occupancy_from_stp(double stp_active, double stp_inactive) {
  if (stp_active + stp_inactive == 0) return 0.0;
  const double tp_i2a = 1.0 - stp_inactive;
  const double tp_a2i = 1.0 - stp_active;
  return clamp01(tp_i2a / (tp_i2a + tp_a2i));
}

/* the shared software libm, exported for the accuracy / bit-parity tests */
double mo_math_log(double x) { return mm_log(x); }
double mo_math_exp(double x) { return mm_exp(x); }
double mo_math_pow(double x, double y) { return mm_pow(x, y); }

/* ============================================================================================
 * Collision word helpers (collision_encoding_impl.hpp:75-242)
 * ========================================================================================== */
static inline uint64_t coll_make(uint64_t idx, unsigned ev) {
  return (idx & MO_INDEX_MASK) | ((uint64_t)ev << MO_EVENT_SHIFT);
}
static inline unsigned coll_event(uint64_t c) { return (unsigned)(c >> MO_EVENT_SHIFT); }
static inline uint64_t coll_index(uint64_t c) { return c & MO_INDEX_MASK; }
static inline int coll_occurred(uint64_t c) { return (coll_event(c) & MO_EV_COLLISION) != 0; }
static inline int coll_occurred_as(uint64_t c, unsigned what) {
  return coll_event(c) == (what | MO_EV_COLLISION);
}
static inline int coll_avoided_as(uint64_t c, unsigned what) {
  return !coll_occurred(c) && coll_event(c) == what;
}
static inline int bound(const uint64_t* epoch, uint64_t i) { return epoch[i] != MO_UNBOUND; }

/* ============================================================================================
 * rank_lefs (simulation.cpp:410-496)
 *
 * The reference sorts the rank arrays by position with cpp-sort's (unstable) split/pdq sort and
 * then re-orders runs of equal positions by binding epoch (rev: ascending, fwd: descending) with
 * a stable insertion sort.  Units that tie on position AND epoch keep whatever order the
 * unstable sort left.  cpp-sort is not available here, so this restatement fixes a total order:
 * (position, epoch rule, position in the incoming rank array).  SURVEY.md H3 documents this as
 * a possible divergence from an official binary (expected ~3e-5 events per chr1 epoch).
 * ========================================================================================== */
typedef struct {
  const uint64_t* pos;
  const uint64_t* epoch;
  const uint64_t* where; /* where[lef] = index in the incoming rank array */
  int epoch_desc;
} rank_cmp_t;

static inline int rank_less(const rank_cmp_t* c, uint64_t a, uint64_t b) {
  if (c->pos[a] != c->pos[b]) return c->pos[a] < c->pos[b];
  if (c->epoch[a] != c->epoch[b])
    return c->epoch_desc ? c->epoch[b] < c->epoch[a] : c->epoch[a] < c->epoch[b];
  return c->where[a] < c->where[b];
}

static void rank_sort_small(const rank_cmp_t* c, uint64_t* v, size_t n, uint64_t* tmp) {
  /* top-down merge sort on the (small) set of displaced entries */
  if (n < 2) return;
  if (n <= 8) {
    for (size_t i = 1; i < n; ++i) {
      const uint64_t x = v[i];
      size_t j = i;
      while (j > 0 && rank_less(c, x, v[j - 1])) {
        v[j] = v[j - 1];
        --j;
      }
      v[j] = x;
    }
    return;
  }
  const size_t h = n / 2;
  rank_sort_small(c, v, h, tmp);
  rank_sort_small(c, v + h, n - h, tmp);
  memcpy(tmp, v, h * sizeof(uint64_t));
  size_t i = 0, j = h, k = 0;
  while (i < h && j < n) v[k++] = rank_less(c, v[j], tmp[i]) ? v[j++] : tmp[i++];
  while (i < h) v[k++] = tmp[i++];
}

static void rank_sort(size_t n, uint64_t* rank, const uint64_t* pos, const uint64_t* epoch,
                      int epoch_desc, uint64_t* scratch /* 3n */) {
  uint64_t* where = scratch;
  uint64_t* kept = scratch + n;
  uint64_t* removed = scratch + 2 * n;
  for (size_t k = 0; k < n; ++k) where[rank[k]] = k;
  const rank_cmp_t c = {pos, epoch, where, epoch_desc};
  /* split: keep a non-decreasing chain; an element that breaks it is removed together with the
   * chain's tail (Levcopoulos-Petersson), so O(n + k log k) on nearly sorted input */
  size_t nk = 0, nr = 0;
  for (size_t k = 0; k < n; ++k) {
    const uint64_t x = rank[k];
    if (nk > 0 && rank_less(&c, x, kept[nk - 1])) {
      removed[nr++] = kept[--nk];
      removed[nr++] = x;
    } else {
      kept[nk++] = x;
    }
  }
  if (nr == 0) return;
  /* `rank` is free to serve as merge scratch here: kept/removed hold every entry */
  rank_sort_small(&c, removed, nr, rank);
  size_t i = 0, j = 0, k = 0;
  while (i < nk && j < nr) rank[k++] = rank_less(&c, removed[j], kept[i]) ? removed[j++] : kept[i++];
  while (i < nk) rank[k++] = kept[i++];
  while (j < nr) rank[k++] = removed[j++];
}

void mo_rank_lefs(size_t n, const uint64_t* rev_pos, const uint64_t* fwd_pos,
                  const uint64_t* epoch, uint64_t* rev_rank, uint64_t* fwd_rank,
                  int init_buffers) {
  if (n == 0) return;
  if (init_buffers) {
    for (size_t i = 0; i < n; ++i) rev_rank[i] = fwd_rank[i] = i;
  }
  uint64_t* scratch = (uint64_t*)malloc(3 * n * sizeof(uint64_t));
  rank_sort(n, rev_rank, rev_pos, epoch, 0, scratch);
  rank_sort(n, fwd_rank, fwd_pos, epoch, 1, scratch);
  free(scratch);
}

/* ============================================================================================
 * Moves (simulation.cpp:272-407)
 * ========================================================================================== */
void mo_adjust_moves(uint64_t start, uint64_t end, size_t n, const uint64_t* rev_pos,
                     const uint64_t* fwd_pos, const uint64_t* epoch, const uint64_t* rev_rank,
                     const uint64_t* fwd_rank, uint64_t* rev_moves, uint64_t* fwd_moves) {
  /* simulation.cpp:359-384: rev units, 3'->5' */
  for (size_t i = n - 1; i > 0; --i) {
    const uint64_t i1 = rev_rank[i - 1];
    const uint64_t i2 = rev_rank[i];
    if (bound(epoch, i1) && bound(epoch, i2)) {
      if (rev_pos[i1] <= start + rev_moves[i1] || rev_pos[i2] <= start + rev_moves[i2]) continue;
      const uint64_t pos1 = rev_pos[i1] - rev_moves[i1];
      const uint64_t pos2 = rev_pos[i2] - rev_moves[i2];
      if (pos2 <= pos1) rev_moves[i1] += (pos1 - pos2) + 1;
    }
  }
  /* simulation.cpp:387-406: fwd units, 5'->3' */
  for (size_t i = 1; i < n; ++i) {
    const uint64_t i1 = fwd_rank[i - 1];
    const uint64_t i2 = fwd_rank[i];
    if (bound(epoch, i1) && bound(epoch, i2)) {
      if (fwd_pos[i1] + fwd_moves[i1] > end - 1 || fwd_pos[i2] + fwd_moves[i2] > end - 1) continue;
      const uint64_t pos1 = fwd_pos[i1] + fwd_moves[i1];
      const uint64_t pos2 = fwd_pos[i2] + fwd_moves[i2];
      if (pos1 >= pos2) fwd_moves[i2] += (pos1 - pos2) + 1;
    }
  }
}

void mo_clamp_moves(uint64_t start, uint64_t end, size_t n, const uint64_t* rev_pos,
                    const uint64_t* fwd_pos, const uint64_t* epoch, uint64_t* rev_moves,
                    uint64_t* fwd_moves) {
  /* simulation.cpp:332-347 */
  for (size_t i = 0; i < n; ++i) {
    if (!bound(epoch, i)) continue;
    rev_moves[i] = MIN(rev_moves[i], rev_pos[i] - start);
    fwd_moves[i] = MIN(fwd_moves[i], end - fwd_pos[i] - 1);
  }
}

static void generate_moves_helper(size_t n, const uint64_t* epoch, uint64_t* moves, double speed,
                                  double std, mo_prng_t* g) {
  /* simulation.cpp:272-297 */
  const uint64_t move_int = (uint64_t)round(speed);
  for (size_t i = 0; i < n; ++i) {
    if (!bound(epoch, i)) {
      moves[i] = 0;
    } else if (std == 0.0) {
      moves[i] = move_int;
    } else {
      moves[i] = (uint64_t)round(fmax(0.0, mo_normal(g, speed, std)));
    }
  }
}

void mo_generate_moves(const mo_params_t* p, uint64_t start, uint64_t end, size_t n,
                       const uint64_t* rev_pos, const uint64_t* fwd_pos, const uint64_t* epoch,
                       const uint64_t* rev_rank, const uint64_t* fwd_rank, uint64_t* rev_moves,
                       uint64_t* fwd_moves, int burnin_completed, mo_prng_t* g, int adjust) {
  /* simulation.cpp:299-330 */
  const double rev_speed =
      (double)(burnin_completed ? p->rev_extrusion_speed : p->rev_extrusion_speed_burnin);
  const double fwd_speed =
      (double)(burnin_completed ? p->fwd_extrusion_speed : p->fwd_extrusion_speed_burnin);
  generate_moves_helper(n, epoch, rev_moves, rev_speed, p->rev_extrusion_speed_std, g);
  generate_moves_helper(n, epoch, fwd_moves, fwd_speed, p->fwd_extrusion_speed_std, g);
  if (adjust)
    mo_adjust_moves(start, end, n, rev_pos, fwd_pos, epoch, rev_rank, fwd_rank, rev_moves,
                    fwd_moves);
  mo_clamp_moves(start, end, n, rev_pos, fwd_pos, epoch, rev_moves, fwd_moves);
}

/* ============================================================================================
 * Collision detection (simulation_detect_collisions.cpp)
 * ========================================================================================== */
void mo_detect_units_at_interval_boundaries(uint64_t start, uint64_t end, size_t n,
                                            const uint64_t* rev_pos, const uint64_t* fwd_pos,
                                            const uint64_t* epoch, const uint64_t* rev_rank,
                                            const uint64_t* fwd_rank, const uint64_t* rev_moves,
                                            const uint64_t* fwd_moves, uint64_t* rev_coll,
                                            uint64_t* fwd_coll, uint64_t* n5_out,
                                            uint64_t* n3_out) {
  /* simulation_detect_collisions.cpp:25-120 */
  uint64_t n5 = 0, n3 = 0;
  const uint64_t first_active_fwd_pos = fwd_pos[fwd_rank[0]];
  uint64_t last_active_rev_pos = 0;
  for (size_t k = n; k-- > 0;) {
    if (bound(epoch, rev_rank[k])) {
      last_active_rev_pos = rev_pos[rev_rank[k]];
      break;
    }
  }
  for (size_t i = 0; i < n; ++i) {
    const uint64_t idx = rev_rank[i];
    const uint64_t pos = rev_pos[idx];
    const uint64_t move = rev_moves[idx];
    if (pos == start) {
      ++n5;
      rev_coll[idx] = coll_make(5, MO_EV_COLLISION | MO_EV_CHROM_BOUNDARY);
    } else if (pos > first_active_fwd_pos) {
      break;
    } else if (pos - move == start) {
      rev_coll[idx] = coll_make(5, MO_EV_COLLISION | MO_EV_CHROM_BOUNDARY);
      ++n5;
      break;
    }
  }
  for (size_t i = n - 1; i > 0; --i) {
    const uint64_t idx = fwd_rank[i];
    const uint64_t pos = fwd_pos[idx];
    const uint64_t move = fwd_moves[idx];
    if (!bound(epoch, idx)) {
      ++n3;
      continue;
    }
    if (pos == end - 1) {
      ++n3;
      fwd_coll[idx] = coll_make(3, MO_EV_COLLISION | MO_EV_CHROM_BOUNDARY);
    } else if (pos < last_active_rev_pos) {
      break;
    } else if (pos + move == end - 1) {
      fwd_coll[idx] = coll_make(3, MO_EV_COLLISION | MO_EV_CHROM_BOUNDARY);
      ++n3;
      break;
    }
  }
  *n5_out = n5;
  *n3_out = n3;
}

/* simulation_impl.hpp:93-101 */
static inline int lef_lef_trial(const mo_params_t* p, mo_prng_t* g) {
  return p->probability_of_extrusion_unit_bypass == 0.0 ||
         mo_bernoulli(g, 1.0 - p->probability_of_extrusion_unit_bypass);
}
static inline int lef_bar_trial(double pblock, mo_prng_t* g) {
  return pblock == 1.0 || mo_bernoulli(g, pblock);
}

void mo_detect_lef_bar_collisions(const mo_params_t* p, size_t n, const uint64_t* rev_pos,
                                  const uint64_t* fwd_pos, const uint64_t* epoch,
                                  const uint64_t* rev_rank, const uint64_t* fwd_rank,
                                  const uint64_t* rev_moves, const uint64_t* fwd_moves,
                                  size_t nb, const uint64_t* bar_pos, const uint8_t* bar_dir,
                                  const uint8_t* bar_active, uint64_t* rev_coll,
                                  uint64_t* fwd_coll, mo_prng_t* g, uint64_t n5, uint64_t n3) {
  /* simulation_detect_collisions.cpp:123-247 */
  size_t j = MIN(n5, n5 - 1);
  uint64_t unit_idx = rev_rank[j];
  uint64_t unit_pos = rev_pos[unit_idx];
  int rev_done = 0;
  for (size_t i = 0; i < nb && !rev_done; ++i) {
    if (!bar_active[i]) continue;
    const double pblock = bar_dir[i] == MO_DIR_REV ? p->lef_bar_major_collision_pblock
                                                   : p->lef_bar_minor_collision_pblock;
    while (unit_pos <= bar_pos[i]) {
      if (++j == n) {
        rev_done = 1;
        break;
      }
      unit_idx = rev_rank[j];
      unit_pos = rev_pos[unit_idx];
    }
    if (rev_done) break;
    if (bound(epoch, unit_idx)) {
      const uint64_t delta = unit_pos - bar_pos[i];
      if (delta > 0 && delta <= rev_moves[unit_idx] && lef_bar_trial(pblock, g)) {
        rev_coll[unit_idx] = coll_make(i, MO_EV_COLLISION | MO_EV_LEF_BAR);
      }
    }
  }

  j = n - MIN(n3, n3 - 1);
  unit_idx = fwd_rank[--j];
  unit_pos = fwd_pos[unit_idx];
  for (size_t i = nb - 1; i != SIZE_MAX; --i) {
    if (!bar_active[i]) continue;
    const double pblock = bar_dir[i] == MO_DIR_FWD ? p->lef_bar_major_collision_pblock
                                                   : p->lef_bar_minor_collision_pblock;
    while (unit_pos >= bar_pos[i]) {
      if (--j == SIZE_MAX) return;
      unit_idx = fwd_rank[j];
      unit_pos = fwd_pos[unit_idx];
    }
    if (bound(epoch, unit_idx)) {
      const uint64_t delta = bar_pos[i] - unit_pos;
      if (delta > 0 && delta <= fwd_moves[unit_idx] && lef_bar_trial(pblock, g)) {
        fwd_coll[unit_idx] = coll_make(i, MO_EV_COLLISION | MO_EV_LEF_BAR);
      }
    }
  }
}

/* simulation.cpp:523-551: (rev landing, fwd landing) */
static void lef_lef_collision_pos(uint64_t rev_p, uint64_t fwd_p, uint64_t rev_move,
                                  uint64_t fwd_move, uint64_t* out_rev, uint64_t* out_fwd) {
  const uint64_t relative_speed = rev_move + fwd_move;
  const double time_to_collision = (double)(rev_p - fwd_p) / (double)relative_speed;
  const uint64_t collision_pos = fwd_p + (uint64_t)round((double)fwd_move * time_to_collision);
  if (collision_pos == fwd_p) {
    *out_rev = collision_pos + 1;
    *out_fwd = collision_pos;
    return;
  }
  *out_rev = collision_pos;
  *out_fwd = collision_pos - 1;
}

/* The reference reads barriers.pos(collision.decode_index()) without checking that the
 * collision is a LEF-BAR one (simulation_detect_collisions.cpp:371, 389; asserted in debug
 * builds only).  A boundary-flagged unit (index 5 / 3) reaching this point makes it read barrier
 * #5 / #3 -- or out of bounds when there are fewer barriers, which is undefined behaviour in
 * the reference and reads as position 0 here. */
static uint64_t stalling_barrier_pos(size_t nb, const uint64_t* bar_pos, uint64_t word) {
  const uint64_t idx = coll_index(word);
  return idx < nb ? bar_pos[idx] : 0;
}

void mo_detect_primary_lef_lef_collisions(const mo_params_t* p, size_t n, const uint64_t* rev_pos,
                                          const uint64_t* fwd_pos, const uint64_t* rev_rank,
                                          const uint64_t* fwd_rank, const uint64_t* rev_moves,
                                          const uint64_t* fwd_moves, size_t nb,
                                          const uint64_t* bar_pos, uint64_t* rev_coll,
                                          uint64_t* fwd_coll, mo_prng_t* g, uint64_t n5,
                                          uint64_t n3) {
  /* simulation_detect_collisions.cpp:250-397 */
  if (n5 == n || n3 == n) return;
  size_t i1 = 0;
  size_t j1 = n5;
  const size_t i2 = n - MIN(n3, n3 - 1);
  const size_t j2 = n;
  for (;;) {
    uint64_t rev_idx = rev_rank[j1];
    uint64_t rev_p = rev_pos[rev_idx];
    uint64_t fwd_idx = fwd_rank[i1];
    uint64_t fwd_p = fwd_pos[fwd_idx];
    while (rev_p <= fwd_p) {
      if (++j1 == j2) return;
      rev_idx = rev_rank[j1];
      rev_p = rev_pos[rev_idx];
    }
    while (fwd_p < rev_p) {
      if (++i1 == i2) return;
      fwd_idx = fwd_rank[i1];
      fwd_p = fwd_pos[fwd_idx];
    }
    fwd_idx = fwd_rank[MIN(i1, i1 - 1)];
    fwd_p = fwd_pos[fwd_idx];

    const uint64_t delta = rev_p - fwd_p;
    if (delta > 0 && delta < rev_moves[rev_idx] + fwd_moves[fwd_idx] && lef_lef_trial(p, g)) {
      const uint64_t rev_move = rev_moves[rev_idx];
      const uint64_t fwd_move = fwd_moves[fwd_idx];
      uint64_t cpos_rev, cpos_fwd;
      lef_lef_collision_pos(rev_p, fwd_p, rev_move, fwd_move, &cpos_rev, &cpos_fwd);
      const int rev_occ = coll_occurred(rev_coll[rev_idx]);
      const int fwd_occ = coll_occurred(fwd_coll[fwd_idx]);
      const unsigned prim = MO_EV_COLLISION | MO_EV_LEF_LEF_PRIMARY;
      if (!rev_occ && !fwd_occ) {
        rev_coll[rev_idx] = coll_make(fwd_idx, prim);
        fwd_coll[fwd_idx] = coll_make(rev_idx, prim);
      } else if (rev_occ && !fwd_occ) {
        const uint64_t barrier_pos = stalling_barrier_pos(nb, bar_pos, rev_coll[rev_idx]);
        if (cpos_fwd > barrier_pos) {
          rev_coll[rev_idx] = coll_make(fwd_idx, prim);
          fwd_coll[fwd_idx] = coll_make(rev_idx, prim);
        } else {
          fwd_coll[fwd_idx] = coll_make(rev_idx, prim);
        }
      } else if (!rev_occ && fwd_occ) {
        const uint64_t barrier_pos = stalling_barrier_pos(nb, bar_pos, fwd_coll[fwd_idx]);
        rev_coll[rev_idx] = coll_make(fwd_idx, prim);
        if (cpos_rev < barrier_pos) fwd_coll[fwd_idx] = coll_make(rev_idx, prim);
      }
    }
  }
}

void mo_correct_moves_for_lef_bar_collisions(size_t n, const uint64_t* rev_pos,
                                             const uint64_t* fwd_pos, const uint64_t* bar_pos,
                                             uint64_t* rev_moves, uint64_t* fwd_moves,
                                             const uint64_t* rev_coll, const uint64_t* fwd_coll) {
  /* simulation_correct_moves.cpp:19-50 */
  for (size_t i = 0; i < n; ++i) {
    if (coll_occurred_as(rev_coll[i], MO_EV_LEF_BAR)) {
      const uint64_t bp = bar_pos[coll_index(rev_coll[i])];
      rev_moves[i] = (rev_pos[i] - bp) - 1;
    }
    if (coll_occurred_as(fwd_coll[i], MO_EV_LEF_BAR)) {
      const uint64_t bp = bar_pos[coll_index(fwd_coll[i])];
      fwd_moves[i] = (bp - fwd_pos[i]) - 1;
    }
  }
}

void mo_correct_moves_for_primary_lef_lef_collisions(size_t n, const uint64_t* rev_pos,
                                                     const uint64_t* fwd_pos,
                                                     const uint64_t* rev_rank,
                                                     const uint64_t* fwd_rank, uint64_t* rev_moves,
                                                     uint64_t* fwd_moves, const uint64_t* rev_coll,
                                                     const uint64_t* fwd_coll) {
  /* simulation_correct_moves.cpp:53-121 */
  for (size_t k = 0; k < n; ++k) {
    const uint64_t rev_idx = rev_rank[k];
    if (coll_occurred_as(rev_coll[rev_idx], MO_EV_LEF_LEF_PRIMARY)) {
      const uint64_t fwd_idx = coll_index(rev_coll[rev_idx]);
      if (coll_occurred_as(fwd_coll[fwd_idx], MO_EV_LEF_LEF_PRIMARY)) {
        uint64_t p1, p2;
        lef_lef_collision_pos(rev_pos[rev_idx], fwd_pos[fwd_idx], rev_moves[rev_idx],
                              fwd_moves[fwd_idx], &p1, &p2);
        rev_moves[rev_idx] = rev_pos[rev_idx] - p1;
        fwd_moves[fwd_idx] = p2 - fwd_pos[fwd_idx];
      } else if (coll_occurred_as(fwd_coll[fwd_idx], MO_EV_LEF_BAR)) {
        rev_moves[rev_idx] = rev_pos[rev_idx] - (fwd_pos[fwd_idx] + fwd_moves[fwd_idx]) - 1;
      }
    }
  }
  for (size_t k = 0; k < n; ++k) {
    const uint64_t fwd_idx = fwd_rank[k];
    if (coll_occurred_as(fwd_coll[fwd_idx], MO_EV_LEF_LEF_PRIMARY)) {
      const uint64_t rev_idx = coll_index(fwd_coll[fwd_idx]);
      if (coll_occurred_as(rev_coll[rev_idx], MO_EV_LEF_BAR)) {
        fwd_moves[fwd_idx] = (rev_pos[rev_idx] - rev_moves[rev_idx]) - fwd_pos[fwd_idx] - 1;
      }
    }
  }
}

void mo_process_secondary_lef_lef_collisions(const mo_params_t* p, size_t n,
                                             const uint64_t* rev_pos, const uint64_t* fwd_pos,
                                             const uint64_t* rev_rank, const uint64_t* fwd_rank,
                                             uint64_t* rev_moves, uint64_t* fwd_moves,
                                             uint64_t* rev_coll, uint64_t* fwd_coll, mo_prng_t* g,
                                             uint64_t n5, uint64_t n3) {
  /* simulation_detect_collisions.cpp:400-515 */
  const unsigned sec = MO_EV_LEF_LEF_SECONDARY;
  for (size_t i = MAX((uint64_t)1, n5); i < n; ++i) {
    const uint64_t idx1 = rev_rank[i - 1];
    if (!coll_occurred(rev_coll[idx1])) continue;
    const uint64_t idx2 = rev_rank[i];
    if (coll_occurred(rev_coll[idx2])) continue;
    const uint64_t pos1 = rev_pos[idx1], pos2 = rev_pos[idx2];
    const uint64_t move1 = rev_moves[idx1];
    if (pos2 - rev_moves[idx2] <= pos1 - move1) {
      if (lef_lef_trial(p, g)) {
        rev_coll[idx2] = coll_make(idx1, MO_EV_COLLISION | sec);
        const uint64_t move = pos2 - (pos1 - move1);
        rev_moves[idx2] = MIN(move, move - 1);
      } else {
        rev_coll[idx2] = coll_make(idx1, sec);
      }
    }
  }
  size_t i = n - MIN(n3, n3 - 1) - 1;
  for (; i > 0; --i) {
    const uint64_t idx2 = fwd_rank[i];
    if (!coll_occurred(fwd_coll[idx2])) continue;
    const uint64_t idx1 = fwd_rank[i - 1];
    if (coll_occurred(fwd_coll[idx1])) continue;
    const uint64_t pos1 = fwd_pos[idx1], pos2 = fwd_pos[idx2];
    const uint64_t move2 = fwd_moves[idx2];
    if (pos1 + fwd_moves[idx1] >= pos2 + move2) {
      if (lef_lef_trial(p, g)) {
        fwd_coll[idx1] = coll_make(idx2, MO_EV_COLLISION | sec);
        const uint64_t move = (pos2 + move2) - pos1;
        fwd_moves[idx1] = MIN(move, move - 1);
      } else {
        fwd_coll[idx1] = coll_make(idx2, sec);
      }
    }
  }
}

static inline void swap64(uint64_t* a, uint64_t* b) {
  const uint64_t t = *a;
  *a = *b;
  *b = t;
}

void mo_fix_secondary_lef_lef_collisions(uint64_t start, uint64_t end, size_t n, uint64_t* rev_pos,
                                         uint64_t* fwd_pos, uint64_t* rev_rank, uint64_t* fwd_rank,
                                         uint64_t* rev_moves, uint64_t* fwd_moves,
                                         uint64_t* rev_coll, uint64_t* fwd_coll, uint64_t n5,
                                         uint64_t n3) {
  /* simulation_detect_collisions.cpp:517-644 */
  const unsigned sec = MO_EV_LEF_LEF_SECONDARY;
  const size_t num_active_fwd_units = n - MIN(n3, n3 - 1);
  for (size_t i = MAX((uint64_t)1, n5); i < n; ++i) {
    const uint64_t idx2 = rev_rank[i];
    if (coll_avoided_as(rev_coll[idx2], sec)) {
      const uint64_t idx1 = rev_rank[i - 1];
      const uint64_t pos1 = rev_pos[idx1] - rev_moves[idx1];
      if (rev_pos[idx2] > pos1 + 1) {
        rev_moves[idx2] = rev_pos[idx2] - (pos1 + 1);
      } else {
        rev_moves[idx2] = 0;
      }
      rev_coll[idx2] = coll_make(idx1, MO_EV_COLLISION | sec);
      const uint64_t p1 = rev_pos[idx1];
      const uint64_t p2 = rev_pos[idx2];
      rev_pos[idx1] = MIN(fwd_pos[idx1], p2);
      rev_pos[idx2] = MIN(fwd_pos[idx2], p1);
      swap64(&rev_coll[idx1], &rev_coll[idx2]);
      swap64(&rev_moves[idx1], &rev_moves[idx2]);
      swap64(&rev_rank[i - 1], &rev_rank[i]);
      const uint64_t a = rev_rank[i - 1], b = rev_rank[i];
      rev_moves[a] = MIN(rev_pos[a] - start, rev_moves[a]);
      rev_moves[b] = MIN(rev_pos[b] - start, rev_moves[b]);
    }
  }
  for (size_t i = 0; i < num_active_fwd_units - 1; ++i) {
    const uint64_t idx1 = fwd_rank[i];
    if (coll_avoided_as(fwd_coll[idx1], sec)) {
      const uint64_t idx2 = fwd_rank[i + 1];
      const uint64_t pos2 = fwd_pos[idx2] + fwd_moves[idx2];
      if (pos2 > fwd_pos[idx1] + 1) {
        fwd_moves[idx1] = pos2 - (fwd_pos[idx1] + 1);
      } else {
        fwd_moves[idx1] = 0;
      }
      fwd_coll[idx1] = coll_make(idx2, MO_EV_COLLISION | sec);
      const uint64_t p1 = fwd_pos[idx1];
      const uint64_t p2 = fwd_pos[idx2];
      fwd_pos[idx1] = MAX(rev_pos[idx1], p2);
      fwd_pos[idx2] = MAX(rev_pos[idx2], p1);
      swap64(&fwd_coll[idx1], &fwd_coll[idx2]);
      swap64(&fwd_moves[idx1], &fwd_moves[idx2]);
      swap64(&fwd_rank[i], &fwd_rank[i + 1]);
      const uint64_t a = fwd_rank[i], b = fwd_rank[i + 1];
      fwd_moves[a] = MIN(end - 1 - fwd_pos[a], fwd_moves[a]);
      fwd_moves[b] = MIN(end - 1 - fwd_pos[b], fwd_moves[b]);
    }
  }
}

void mo_process_collisions(const mo_params_t* p, uint64_t start, uint64_t end, size_t n,
                           uint64_t* rev_pos, uint64_t* fwd_pos, const uint64_t* epoch,
                           uint64_t* rev_rank, uint64_t* fwd_rank, uint64_t* rev_moves,
                           uint64_t* fwd_moves, size_t nb, const uint64_t* bar_pos,
                           const uint8_t* bar_dir, const uint8_t* bar_active, uint64_t* rev_coll,
                           uint64_t* fwd_coll, mo_prng_t* g, int with_fix) {
  /* simulation.cpp:763-793 */
  uint64_t n5, n3;
  mo_detect_units_at_interval_boundaries(start, end, n, rev_pos, fwd_pos, epoch, rev_rank,
                                         fwd_rank, rev_moves, fwd_moves, rev_coll, fwd_coll, &n5,
                                         &n3);
  mo_detect_lef_bar_collisions(p, n, rev_pos, fwd_pos, epoch, rev_rank, fwd_rank, rev_moves,
                               fwd_moves, nb, bar_pos, bar_dir, bar_active, rev_coll, fwd_coll, g,
                               n5, n3);
  mo_detect_primary_lef_lef_collisions(p, n, rev_pos, fwd_pos, rev_rank, fwd_rank, rev_moves,
                                       fwd_moves, nb, bar_pos, rev_coll, fwd_coll, g, n5, n3);
  mo_correct_moves_for_lef_bar_collisions(n, rev_pos, fwd_pos, bar_pos, rev_moves, fwd_moves,
                                          rev_coll, fwd_coll);
  mo_correct_moves_for_primary_lef_lef_collisions(n, rev_pos, fwd_pos, rev_rank, fwd_rank,
                                                  rev_moves, fwd_moves, rev_coll, fwd_coll);
  mo_process_secondary_lef_lef_collisions(p, n, rev_pos, fwd_pos, rev_rank, fwd_rank, rev_moves,
                                          fwd_moves, rev_coll, fwd_coll, g, n5, n3);
  if (with_fix)
    mo_fix_secondary_lef_lef_collisions(start, end, n, rev_pos, fwd_pos, rev_rank, fwd_rank,
                                        rev_moves, fwd_moves, rev_coll, fwd_coll, n5, n3);
}

/* ============================================================================================
 * Whole cell (simulation.cpp:896-986)
 * ========================================================================================== */
typedef struct {
  const mo_params_t* p;
  uint64_t start, end;
  size_t nb;
  const uint64_t* bar_pos;
  const uint8_t* bar_dir;
  const double* bar_stp_active;
  const double* bar_stp_inactive;
  uint8_t* bar_active;
  size_t num_lefs, num_active;
  uint64_t *rev_pos, *fwd_pos, *epoch, *rev_rank, *fwd_rank, *rev_moves, *fwd_moves, *rev_coll,
      *fwd_coll, *scratch;
  uint32_t* contacts;
  uint64_t nrows, ncols;
  uint64_t* missed;
  uint64_t* occupancy;
  mo_prng_t g;
  /* burn-in history (two deque<double> of capacity burnin_history_length) */
  double *cfx_buff, *avg_buff;
  size_t hist_len;
} cell_t;

static void matrix_increment(cell_t* s, uint64_t row, uint64_t col) {
  /* contact_matrix_internal_impl.hpp:19-42, contact_matrix_dense_safe_impl.hpp:55-68 */
  uint64_t i, j;
  if (row > col) {
    i = row - col;
    j = row;
  } else {
    i = col - row;
    j = col;
  }
  if (i >= s->nrows) {
    __atomic_fetch_add(s->missed, 1, __ATOMIC_RELAXED);
    return;
  }
  __atomic_fetch_add(&s->contacts[j * s->nrows + i], 1u, __ATOMIC_RELAXED);
}

/* stats::mean / stats::sum_of_squared_deviations / stats::variance / stats::standard_dev over the
 * loop sizes (stats/descriptive_impl.hpp:22-31, 63-101): std::accumulate in double, left to
 * right, population variance. */
void mo_loop_size_stats(size_t n, const uint64_t* rev_pos, const uint64_t* fwd_pos, double* avg_out,
                        double* ssd_out, double* var_out, double* std_out) {
  double acc = 0.0;
  for (size_t i = 0; i < n; ++i) acc = acc + (double)(fwd_pos[i] - rev_pos[i]);
  const double avg = acc / (double)n;
  double ssd = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const double d = (double)(fwd_pos[i] - rev_pos[i]) - avg;
    ssd = ssd + (d * d);
  }
  if (avg_out) *avg_out = avg;
  if (ssd_out) *ssd_out = ssd;
  if (var_out) *var_out = ssd / (double)n;
  if (std_out) *std_out = sqrt(ssd / (double)n);
}

/* ContactMatrixDense::increment on a bare band buffer (unit hook for the reference's
 * contact-matrix tests); same arithmetic as matrix_increment above */
void mo_matrix_increment(uint32_t* contacts, uint64_t nrows, uint64_t ncols, uint64_t row,
                         uint64_t col, uint64_t* missed) {
  cell_t s;
  memset(&s, 0, sizeof(s));
  s.contacts = contacts;
  s.nrows = nrows;
  s.ncols = ncols;
  s.missed = missed;
  matrix_increment(&s, row, col);
}

/* Collision<> word and its predicates (collision_encoding_impl.hpp:75-242); bit layout of the
 * result as in include/modle_hip.h (MODLE_HIP_UNIT_COLLISION_WORDS) */
uint64_t mo_collision_word(uint64_t idx, unsigned ev) { return coll_make(idx, ev); }
unsigned mo_collision_predicates(uint64_t w) {
  static const unsigned kinds[4] = {MO_EV_CHROM_BOUNDARY, MO_EV_LEF_BAR, MO_EV_LEF_LEF_PRIMARY,
                                    MO_EV_LEF_LEF_SECONDARY};
  unsigned f = (coll_occurred(w) ? 1u : 0u) | ((!coll_occurred(w) && w != 0) ? 2u : 0u);
  for (unsigned k = 0; k < 4; ++k) {
    f |= coll_occurred_as(w, kinds[k]) ? (4u << k) : 0u;
    f |= coll_avoided_as(w, kinds[k]) ? (64u << k) : 0u;
  }
  return f;
}

static void compute_loop_size_stats(cell_t* s) {
  /* simulation.cpp:795-819 + stats/descriptive_impl.hpp:22-31, 63-101 */
  const size_t n = s->num_active;
  const size_t cap = s->p->burnin_history_length;
  if (n == 0) {
    s->hist_len = 0;
    return;
  }
  double avg, std;
  mo_loop_size_stats(n, s->rev_pos, s->fwd_pos, &avg, NULL, NULL, &std);
  if (s->hist_len == cap) {
    memmove(s->avg_buff, s->avg_buff + 1, (cap - 1) * sizeof(double));
    memmove(s->cfx_buff, s->cfx_buff + 1, (cap - 1) * sizeof(double));
    --s->hist_len;
  }
  s->avg_buff[s->hist_len] = avg;
  s->cfx_buff[s->hist_len] = std / avg;
  ++s->hist_len;
}

static double window_mean(const double* v, size_t w) {
  double acc = 0.0;
  for (size_t i = 0; i < w; ++i) acc = acc + v[i];
  return acc / (double)w;
}

static int series_is_stable(const double* buf, size_t cap, size_t w) {
  size_t n = 0;
  for (size_t j = 0; j + w + 1 < cap; ++j) {
    const double n1 = window_mean(buf + j, w);
    const double n2 = window_mean(buf + j + 1, w);
    n += (size_t)(n1 > n2);
  }
  const double r = (double)n / (double)(cap - w - n);
  return r >= 0.95 && r <= 1.05;
}

static int evaluate_burnin(const cell_t* s) {
  /* simulation.cpp:821-864 */
  const size_t cap = s->p->burnin_history_length;
  const size_t w = s->p->burnin_smoothing_window_size;
  if (s->hist_len != cap) return 0;
  if (!series_is_stable(s->cfx_buff, cap, w)) return 0;
  return series_is_stable(s->avg_buff, cap, w);
}

/* Simulation::select_and_bind_lefs (simulation.cpp:988-993): select_lefs_to_bind (mask = the
 * released LEFs) + bind_lefs (simulation_impl.hpp:30-91) + rank_lefs.  `scratch`: 3 n words. */
void mo_select_and_bind_lefs(uint64_t start, uint64_t end, size_t n, uint64_t* rev_pos,
                             uint64_t* fwd_pos, uint64_t* epoch, uint64_t* rev_rank,
                             uint64_t* fwd_rank, uint64_t epoch_now, mo_prng_t* g,
                             uint64_t* scratch) {
  for (size_t i = 0; i < n; ++i) {
    if (!bound(epoch, i)) {
      const uint64_t pos = mo_uniform_int(g, start, end - 1);
      rev_pos[i] = pos;
      fwd_pos[i] = pos;
      epoch[i] = epoch_now;
    }
  }
  rank_sort(n, rev_rank, rev_pos, epoch, 0, scratch);
  rank_sort(n, fwd_rank, fwd_pos, epoch, 1, scratch);
}

static void bind_lefs(cell_t* s, uint64_t epoch_now) {
  mo_select_and_bind_lefs(s->start, s->end, s->num_active, s->rev_pos, s->fwd_pos, s->epoch,
                          s->rev_rank, s->fwd_rank, epoch_now, &s->g, s->scratch);
}

static inline int lef_within_bound(const cell_t* s, size_t i, uint64_t lo, uint64_t hi) {
  /* register_contacts.cpp:23-29 */
  return s->rev_pos[i] > lo && s->rev_pos[i] < hi && s->fwd_pos[i] > lo && s->fwd_pos[i] < hi;
}

static inline int sample_positions(cell_t* s, uint64_t lo, uint64_t hi, double* p1, double* p2) {
  /* register_contacts.cpp:47-70, 138-147: returns 1 when the event yields a usable pair */
  const size_t n = s->num_active;
  const size_t i = (size_t)mo_uniform_int(&s->g, 0, n - 1);
  if (!(bound(s->epoch, i) && lef_within_bound(s, i, lo, hi))) return 0;
  const mo_params_t* p = s->p;
  const int noisify = (p->contact_sampling_strategy & MO_CS_NOISIFY) != 0;
  const double n1 =
      noisify ? mo_genextreme(&s->g, p->genextreme_mu, p->genextreme_sigma, p->genextreme_xi) : 0.0;
  const double a = (double)s->rev_pos[i] - n1;
  const double n2 =
      noisify ? mo_genextreme(&s->g, p->genextreme_mu, p->genextreme_sigma, p->genextreme_xi) : 0.0;
  const double b = (double)s->fwd_pos[i] + n2;
  /* std::minmax({a, b}) */
  *p1 = b < a ? b : a;
  *p2 = b < a ? a : b;
  const double lo_ = (double)lo, hi_ = (double)hi;
  return *p1 >= lo_ && *p2 >= lo_ && *p1 < hi_ && *p2 < hi_;
}

static uint64_t sample_and_register_contacts(cell_t* s, uint64_t num_events,
                                             uint64_t num_target_contacts, uint64_t num_contacts,
                                             uint64_t* events_done) {
  /* register_contacts.cpp:93-232; returns the number of contacts registered */
  const mo_params_t* p = s->p;
  if (p->target_contact_density > 0.0) num_events = MIN(num_events, num_target_contacts - num_contacts);
  if (num_events == 0) return 0;
  *events_done += num_events;

  uint64_t n_loop;
  if (p->tad_to_loop_contact_ratio == 0) {
    n_loop = num_events;
  } else if (!isfinite(p->tad_to_loop_contact_ratio)) {
    n_loop = 0;
  } else {
    const double prob_loop = 1.0 / (p->tad_to_loop_contact_ratio + 1.0);
    n_loop = (uint64_t)mo_binomial(&s->g, (int64_t)num_events, prob_loop);
  }
  const uint64_t n_tad = num_events - n_loop;
  const uint64_t lo = s->start + 1, hi = s->end - 1;
  uint64_t registered = 0;
  double p1, p2;
  for (uint64_t e = 0; e < n_loop; ++e) {
    if (!sample_positions(s, lo, hi, &p1, &p2)) continue;
    matrix_increment(s, ((uint64_t)p1 - lo) / p->bin_size, ((uint64_t)p2 - lo) / p->bin_size);
    ++registered;
  }
  for (uint64_t e = 0; e < n_tad; ++e) {
    if (!sample_positions(s, lo, hi, &p1, &p2)) continue;
    const uint64_t p11 = mo_uniform_int(&s->g, (uint64_t)p1, (uint64_t)p2);
    const uint64_t p22 = mo_uniform_int(&s->g, (uint64_t)p1, (uint64_t)p2);
    matrix_increment(s, (p11 - lo) / p->bin_size, (p22 - lo) / p->bin_size);
    ++registered;
  }
  if (p->track_1d_lef_position) {
    for (uint64_t e = 0; e < num_events; ++e) {
      if (!sample_positions(s, lo, hi, &p1, &p2)) continue;
      if (s->occupancy) {
        __atomic_fetch_add(&s->occupancy[((uint64_t)p1 - lo) / p->bin_size], 1, __ATOMIC_RELAXED);
        __atomic_fetch_add(&s->occupancy[((uint64_t)p2 - lo) / p->bin_size], 1, __ATOMIC_RELAXED);
      }
    }
  }
  return registered;
}

static void release_lefs(cell_t* s, int burnin_completed) {
  /* simulation.cpp:553-601 */
  const mo_params_t* p = s->p;
  const double base = burnin_completed ? p->prob_of_lef_release : p->prob_of_lef_release_burnin;
  for (size_t i = 0; i < s->num_active; ++i) {
    if (!bound(s->epoch, i)) continue;
    int hard = 0;
    if (coll_occurred_as(s->rev_coll[i], MO_EV_LEF_BAR))
      hard += s->bar_dir[coll_index(s->rev_coll[i])] == MO_DIR_REV;
    if (coll_occurred_as(s->fwd_coll[i], MO_EV_LEF_BAR))
      hard += s->bar_dir[coll_index(s->fwd_coll[i])] == MO_DIR_FWD;
    const double affinity = hard == 0   ? 1.0
                            : hard == 1 ? 1.0 / p->soft_stall_lef_stability_multiplier
                                        : 1.0 / p->hard_stall_lef_stability_multiplier;
    if (mo_bernoulli(&s->g, affinity * base)) {
      s->rev_pos[i] = MO_UNBOUND;
      s->fwd_pos[i] = MO_UNBOUND;
      s->epoch[i] = MO_UNBOUND;
    }
  }
}

/* ExtrusionBarriers::sort (extrusion_barriers.cpp:237-257): index sort by position, then the
 * four arrays are permuted.  The reference uses an unstable pdq_sort; equal positions keep their
 * input order here (a documented choice, like the device's). */
typedef struct {
  uint64_t pos;
  size_t idx;
} bar_key_t;
static int bar_key_cmp(const void* a, const void* b) {
  const bar_key_t *x = (const bar_key_t*)a, *y = (const bar_key_t*)b;
  if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
void mo_sort_barriers(size_t nb, uint64_t* pos, uint8_t* dir, double* stp_active,
                      double* stp_inactive) {
  if (nb < 2) return;
  bar_key_t* k = (bar_key_t*)malloc(nb * sizeof(bar_key_t));
  uint64_t* tp = (uint64_t*)malloc(nb * sizeof(uint64_t));
  uint8_t* td = (uint8_t*)malloc(nb);
  double* ta = (double*)malloc(nb * sizeof(double));
  double* ti = (double*)malloc(nb * sizeof(double));
  for (size_t i = 0; i < nb; ++i) {
    k[i].pos = pos[i];
    k[i].idx = i;
  }
  qsort(k, nb, sizeof(bar_key_t), bar_key_cmp);
  for (size_t i = 0; i < nb; ++i) {
    tp[i] = pos[k[i].idx];
    td[i] = dir[k[i].idx];
    ta[i] = stp_active[k[i].idx];
    ti[i] = stp_inactive[k[i].idx];
  }
  memcpy(pos, tp, nb * sizeof(uint64_t));
  memcpy(dir, td, nb);
  memcpy(stp_active, ta, nb * sizeof(double));
  memcpy(stp_inactive, ti, nb * sizeof(double));
  free(k);
  free(tp);
  free(td);
  free(ta);
  free(ti);
}

static int simulate_cell_sorted(const mo_params_t* p, uint64_t start, uint64_t end, size_t nb,
                                const uint64_t* bar_pos, const uint8_t* bar_dir,
                                const double* bar_stp_active, const double* bar_stp_inactive,
                                const mo_task_t* task, uint32_t* contacts, uint64_t nrows,
                                uint64_t ncols, uint64_t* missed, uint64_t* occupancy,
                                mo_cell_result_t* res);

/* Model-internal-state log (Simulation::dump_stats, simulation.cpp:995-1056; called after
 * extrude and before release_lefs, :969-975): the sink of the calling thread, 10 words per
 * record in the layout of MODLE_HIP_STATE_LOG_WORDS (include/modle_hip.h). */
static __thread uint64_t* tl_state_log = NULL;
static __thread size_t tl_state_log_cap = 0, tl_state_log_n = 0;

static void dump_stats(uint64_t epoch, int burnin, size_t n, const uint64_t* rev_pos,
                       const uint64_t* fwd_pos, const uint64_t* epochs, const uint64_t* rev_coll,
                       const uint64_t* fwd_coll, size_t nb, const uint8_t* bar_active) {
  if (tl_state_log == NULL || tl_state_log_n >= tl_state_log_cap) return;
  uint64_t occ = 0, st_rev = 0, st_fwd = 0, st_both = 0, n_bar = 0, n_prim = 0, n_sec = 0, loops = 0;
  for (size_t i = 0; i < nb; ++i) occ += bar_active[i] != 0;
  for (size_t i = 0; i < n; ++i) {
    const int r = coll_occurred(rev_coll[i]), f = coll_occurred(fwd_coll[i]);
    st_rev += (uint64_t)r;
    st_fwd += (uint64_t)f;
    st_both += (uint64_t)(r && f);
    n_bar += (uint64_t)coll_occurred_as(rev_coll[i], MO_EV_LEF_BAR) + (uint64_t)coll_occurred_as(fwd_coll[i], MO_EV_LEF_BAR);
    n_prim += (uint64_t)coll_occurred_as(rev_coll[i], MO_EV_LEF_LEF_PRIMARY) +
              (uint64_t)coll_occurred_as(fwd_coll[i], MO_EV_LEF_LEF_PRIMARY);
    n_sec += (uint64_t)coll_occurred_as(rev_coll[i], MO_EV_LEF_LEF_SECONDARY) +
             (uint64_t)coll_occurred_as(fwd_coll[i], MO_EV_LEF_LEF_SECONDARY);
    if (bound(epochs, i)) loops += fwd_pos[i] - rev_pos[i];
  }
  uint64_t* rec = tl_state_log + 10 * tl_state_log_n++;
  rec[0] = epoch | (burnin ? (UINT64_C(1) << 63) : 0);
  rec[1] = occ;
  rec[2] = n;
  rec[3] = st_rev;
  rec[4] = st_fwd;
  rec[5] = st_both;
  rec[6] = n_bar;
  rec[7] = n_prim;
  rec[8] = n_sec;
  rec[9] = loops;
}

/* one cell with its internal-state log: `log` holds cap records; returns the number written */
size_t mo_simulate_cell_with_state_log(const mo_params_t* p, uint64_t start, uint64_t end, size_t nb,
                                       const uint64_t* bar_pos, const uint8_t* bar_dir,
                                       const double* bar_stp_active, const double* bar_stp_inactive,
                                       const mo_task_t* task, uint32_t* contacts, uint64_t nrows,
                                       uint64_t ncols, uint64_t* missed, uint64_t* occupancy,
                                       mo_cell_result_t* res, uint64_t* log, size_t cap) {
  tl_state_log = log;
  tl_state_log_cap = cap;
  tl_state_log_n = 0;
  (void)mo_simulate_cell(p, start, end, nb, bar_pos, bar_dir, bar_stp_active, bar_stp_inactive, task,
                         contacts, nrows, ncols, missed, occupancy, res);
  tl_state_log = NULL;
  return tl_state_log_n;
}

/* State::operator=(const Task&) copies the interval's barriers and sorts them for every task
 * (simulation.cpp:741-761); the copy is only made here when the input is not sorted already. */
int mo_simulate_cell(const mo_params_t* p, uint64_t start, uint64_t end, size_t nb,
                     const uint64_t* bar_pos, const uint8_t* bar_dir, const double* bar_stp_active,
                     const double* bar_stp_inactive, const mo_task_t* task, uint32_t* contacts,
                     uint64_t nrows, uint64_t ncols, uint64_t* missed, uint64_t* occupancy,
                     mo_cell_result_t* res) {
  int sorted = 1;
  for (size_t i = 1; i < nb && sorted; ++i) sorted = bar_pos[i - 1] <= bar_pos[i];
  if (sorted)
    return simulate_cell_sorted(p, start, end, nb, bar_pos, bar_dir, bar_stp_active,
                                bar_stp_inactive, task, contacts, nrows, ncols, missed, occupancy,
                                res);
  uint64_t* sp = (uint64_t*)malloc(nb * sizeof(uint64_t));
  uint8_t* sd = (uint8_t*)malloc(nb);
  double* sa = (double*)malloc(nb * sizeof(double));
  double* si = (double*)malloc(nb * sizeof(double));
  memcpy(sp, bar_pos, nb * sizeof(uint64_t));
  memcpy(sd, bar_dir, nb);
  memcpy(sa, bar_stp_active, nb * sizeof(double));
  memcpy(si, bar_stp_inactive, nb * sizeof(double));
  mo_sort_barriers(nb, sp, sd, sa, si);
  const int rc = simulate_cell_sorted(p, start, end, nb, sp, sd, sa, si, task, contacts, nrows,
                                      ncols, missed, occupancy, res);
  free(sp);
  free(sd);
  free(sa);
  free(si);
  return rc;
}

static int simulate_cell_sorted(const mo_params_t* p, uint64_t start, uint64_t end, size_t nb,
                                const uint64_t* bar_pos, const uint8_t* bar_dir,
                                const double* bar_stp_active, const double* bar_stp_inactive,
                                const mo_task_t* task, uint32_t* contacts, uint64_t nrows,
                                uint64_t ncols, uint64_t* missed, uint64_t* occupancy,
                                mo_cell_result_t* res) {
  cell_t s;
  memset(&s, 0, sizeof(s));
  s.p = p;
  s.start = start;
  s.end = end;
  s.nb = nb;
  s.bar_pos = bar_pos;
  s.bar_dir = bar_dir;
  s.bar_stp_active = bar_stp_active;
  s.bar_stp_inactive = bar_stp_inactive;
  s.num_lefs = (size_t)task->num_lefs;
  s.contacts = contacts;
  s.nrows = nrows;
  s.ncols = ncols;
  s.missed = missed;
  s.occupancy = occupancy;
  memcpy(s.g.s, task->prng, sizeof(s.g.s));
  s.g.count = 0;

  const size_t L = s.num_lefs;
  const size_t cap = p->burnin_history_length;
  uint64_t* mem = (uint64_t*)malloc((12 * L + 1) * sizeof(uint64_t));
  s.bar_active = (uint8_t*)malloc(nb + 1);
  s.cfx_buff = (double*)malloc((2 * cap + 2) * sizeof(double));
  if (!mem || !s.bar_active || !s.cfx_buff) {
    free(mem);
    free(s.bar_active);
    free(s.cfx_buff);
    return -1;
  }
  s.avg_buff = s.cfx_buff + cap + 1;
  s.rev_pos = mem;
  s.fwd_pos = mem + L;
  s.epoch = mem + 2 * L;
  s.rev_rank = mem + 3 * L;
  s.fwd_rank = mem + 4 * L;
  s.rev_moves = mem + 5 * L;
  s.fwd_moves = mem + 6 * L;
  s.rev_coll = mem + 7 * L;
  s.fwd_coll = mem + 8 * L;
  s.scratch = mem + 9 * L;
  /* State::reset_buffers (simulation.cpp:617-627) */
  for (size_t i = 0; i < L; ++i) {
    s.rev_pos[i] = s.fwd_pos[i] = s.epoch[i] = MO_UNBOUND;
    s.rev_rank[i] = s.fwd_rank[i] = i;
    s.rev_moves[i] = s.fwd_moves[i] = 0;
    s.rev_coll[i] = s.fwd_coll[i] = 0;
  }

  uint64_t epoch = 0, num_burnin_epochs = 0, num_contacts = 0;
  uint64_t sum_active = 0, events_done = 0, sim_epochs = 0;
  int burnin_completed = 0;
  const double lef_binding_rate_burnin =
      (double)s.num_lefs / (double)p->burnin_target_epochs_for_lef_activation;
  const uint64_t sampling_events_per_epoch = mo_compute_contacts_per_epoch(p, s.num_lefs);

  /* ExtrusionBarriers::init_states (extrusion_barriers.cpp:219-230) */
  for (size_t i = 0; i < nb; ++i) {
    const double occ = mo_occupancy_from_stp(bar_stp_active[i], bar_stp_inactive[i]);
    s.bar_active[i] = (uint8_t)mo_bernoulli(&s.g, occ);
  }
  if (p->skip_burnin) {
    s.num_active = s.num_lefs;
    burnin_completed = 1;
  }

  for (;; ++epoch) {
    /* stop_condition (simulation.cpp:925-931) */
    if (p->target_contact_density >= 0) {
      if (num_contacts >= task->num_target_contacts) break;
    } else if (epoch - num_burnin_epochs >= task->num_target_epochs) {
      break;
    }

    if (!burnin_completed) {
      /* run_burnin (simulation.cpp:866-894) */
      do {
        ++num_burnin_epochs;
        if (s.num_active != s.num_lefs) {
          const uint64_t k = mo_poisson(&s.g, lef_binding_rate_burnin);
          s.num_active = MIN(s.num_active + k, s.num_lefs);
        } else {
          compute_loop_size_stats(&s);
          burnin_completed = evaluate_burnin(&s);
          burnin_completed &= epoch > p->min_burnin_epochs;
          if (!burnin_completed && epoch >= p->max_burnin_epochs) {
            burnin_completed = 1;
            s.num_active = s.num_lefs;
          }
        }
      } while (s.num_active == 0);
    }

    bind_lefs(&s, epoch);
    if (burnin_completed) {
      num_contacts += sample_and_register_contacts(&s, sampling_events_per_epoch,
                                                   task->num_target_contacts, num_contacts,
                                                   &events_done);
      if (task->num_target_contacts != 0 && num_contacts >= task->num_target_contacts) break;
    }

    const size_t n = s.num_active;
    sum_active += n;
    ++sim_epochs;
    mo_generate_moves(p, start, end, n, s.rev_pos, s.fwd_pos, s.epoch, s.rev_rank, s.fwd_rank,
                      s.rev_moves, s.fwd_moves, burnin_completed, &s.g, 1);

    /* ExtrusionBarriers::next_state (extrusion_barriers.cpp:145-161) */
    for (size_t i = 0; i < nb; ++i) {
      const double u = mo_canonical(&s.g);
      if (!s.bar_active[i] && u > bar_stp_inactive[i]) {
        s.bar_active[i] = 1;
      } else if (s.bar_active[i] && u > bar_stp_active[i]) {
        s.bar_active[i] = 0;
      }
    }
    memset(s.rev_coll, 0, n * sizeof(uint64_t));
    memset(s.fwd_coll, 0, n * sizeof(uint64_t));
    mo_process_collisions(p, start, end, n, s.rev_pos, s.fwd_pos, s.epoch, s.rev_rank, s.fwd_rank,
                          s.rev_moves, s.fwd_moves, nb, bar_pos, bar_dir, s.bar_active, s.rev_coll,
                          s.fwd_coll, &s.g, 1);
    /* extrude (simulation.cpp:498-521) */
    for (size_t i = 0; i < n; ++i) {
      if (!bound(s.epoch, i)) continue;
      s.rev_pos[i] -= s.rev_moves[i];
      s.fwd_pos[i] += s.fwd_moves[i];
    }
    dump_stats(epoch, !burnin_completed, n, s.rev_pos, s.fwd_pos, s.epoch, s.rev_coll, s.fwd_coll, nb,
               s.bar_active);
    release_lefs(&s, burnin_completed);
  }

  if (res) {
    res->epochs = epoch;
    res->burnin_epochs = num_burnin_epochs;
    res->num_contacts = num_contacts;
    res->raws_consumed = s.g.count;
    memcpy(res->prng_final, s.g.s, sizeof(s.g.s));
    res->sum_active_lefs = sum_active;
    res->sampling_events = events_done;
    res->sim_epochs = sim_epochs;
  }
  free(mem);
  free(s.bar_active);
  free(s.cfx_buff);
  return 0;
}

/* ---- multi-threaded driver: one worker per host thread over a shared task cursor, the way
 * simulate_worker drains the task queue (scheduler_simulate.cpp:190-271) ------------------- */
typedef struct {
  const mo_params_t* p;
  uint64_t start, end;
  size_t nb;
  const uint64_t* bar_pos;
  const uint8_t* bar_dir;
  const double *bar_stp_active, *bar_stp_inactive;
  const mo_task_t* tasks;
  size_t n_tasks;
  uint32_t* contacts;
  uint64_t nrows, ncols;
  uint64_t* missed;
  uint64_t* occupancy;
  mo_cell_result_t* results;
  size_t cursor;
  int err;
} job_t;

static void* worker_main(void* arg) {
  job_t* j = (job_t*)arg;
  for (;;) {
    const size_t t = __atomic_fetch_add(&j->cursor, 1, __ATOMIC_RELAXED);
    if (t >= j->n_tasks) return NULL;
    const int rc = mo_simulate_cell(j->p, j->start, j->end, j->nb, j->bar_pos, j->bar_dir,
                                    j->bar_stp_active, j->bar_stp_inactive, &j->tasks[t],
                                    j->contacts, j->nrows, j->ncols, j->missed, j->occupancy,
                                    j->results ? &j->results[t] : NULL);
    if (rc != 0) __atomic_store_n(&j->err, rc, __ATOMIC_RELAXED);
  }
}

int mo_simulate_interval(const mo_params_t* p, uint64_t start, uint64_t end, size_t nb,
                         const uint64_t* bar_pos, const uint8_t* bar_dir,
                         const double* bar_stp_active, const double* bar_stp_inactive,
                         const mo_task_t* tasks, size_t n_tasks, uint32_t* contacts,
                         uint64_t nrows, uint64_t ncols, uint64_t* missed, uint64_t* occupancy,
                         mo_cell_result_t* results, int nthreads) {
  job_t j = {p,        start, end,   nb,     bar_pos,   bar_dir, bar_stp_active, bar_stp_inactive,
             tasks,    n_tasks, contacts, nrows, ncols, missed,  occupancy,      results,
             0,        0};
  if (nthreads < 1) nthreads = 1;
  if ((size_t)nthreads > n_tasks) nthreads = (int)(n_tasks ? n_tasks : 1);
  if (nthreads == 1) {
    worker_main(&j);
    return j.err;
  }
  pthread_t* th = (pthread_t*)malloc((size_t)nthreads * sizeof(pthread_t));
  for (int i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, worker_main, &j);
  for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
  free(th);
  return j.err;
}

/* modle_math.h -- software log / exp / pow in IEEE double precision, shared by the device code
 * (sim_rng.h through wave::f_log / f_exp / f_pow), the CPU lane emulator and the CPU oracle.
 *
 * Why: the reference calls the platform libm (glibc) in rejection tests (ziggurat wedges and
 * tails, Poisson PTRD, binomial BTRD / inversion) and in the GEV noise of contact sampling
 * (genextreme_value_distribution.hpp:87-105).  No two libm implementations agree in the last
 * bit, and one flipped comparison changes the rest of a cell's PRNG stream.  With ONE
 * implementation compiled into both sides -- only +, -, *, / and integer operations on the bit
 * patterns, no fused multiply-add (every build uses -ffp-contract=off), no library call -- every
 * libm-dependent decision is identical on CPU and GPU by construction.
 *
 * Method (own design; constants from tools/gen_math_tables.py, computed with mpmath):
 *   log:  x = 2^k z, z in [sqrt(2)/2, sqrt(2)); table of 128 centres c with log(c) as a
 *         double-double; r = (z - c) / c as a double-double (Dekker products); log1p(r) with the
 *         r and r^2/2 terms in double-double and the tail r^3/3 - ... + r^9/9 in double.  The
 *         result is a double-double (about 2^-65 relative), rounded once for log().
 *   exp:  e = hi + lo; k = round(hi * 128 / ln 2); r = hi - k ln2/128 + lo; 2^(k/128) from a table
 *         of 128 double-doubles; expm1(r) by a degree-6 polynomial.
 *   pow:  exp(y * log(x)) with the logarithm and the product carried as double-doubles.
 * Errors stay below one ulp (tests/test_modle_math.py checks against mpmath).  Plain C99 / C++17.
 */
#ifndef MODLE_MATH_H
#define MODLE_MATH_H

#include <stdint.h>

#ifndef MM_FN
#define MM_FN static inline
#endif
#ifndef MM_TABLE
#define MM_TABLE static const
#endif

#include "modle_math_tables.h"

typedef struct {
  double hi, lo;
} mm_dd;

MM_FN uint64_t mm_bits(double x) {
  union {
    double d;
    uint64_t u;
  } v;
  v.d = x;
  return v.u;
}
MM_FN double mm_from_bits(uint64_t u) {
  union {
    double d;
    uint64_t u;
  } v;
  v.u = u;
  return v.d;
}

/* error-free transformations (Knuth / Dekker / Veltkamp); no FMA */
MM_FN mm_dd mm_two_sum(double a, double b) {
  mm_dd r;
  r.hi = a + b;
  const double bb = r.hi - a;
  r.lo = (a - (r.hi - bb)) + (b - bb);
  return r;
}
MM_FN mm_dd mm_fast_two_sum(double a, double b) { /* |a| >= |b| */
  mm_dd r;
  r.hi = a + b;
  r.lo = b - (r.hi - a);
  return r;
}
MM_FN mm_dd mm_split(double a) {
  mm_dd r;
  const double c = 134217729.0 * a; /* 2^27 + 1 */
  r.hi = c - (c - a);
  r.lo = a - r.hi;
  return r;
}
MM_FN mm_dd mm_two_prod(double a, double b) {
  mm_dd r;
  r.hi = a * b;
  const mm_dd x = mm_split(a), y = mm_split(b);
  r.lo = ((x.hi * y.hi - r.hi) + x.hi * y.lo + x.lo * y.hi) + x.lo * y.lo;
  return r;
}

#define MM_INF (mm_from_bits(0x7FF0000000000000ull))
#define MM_NAN (mm_from_bits(0x7FF8000000000000ull))

/* log(x) as a double-double for finite x > 0 */
MM_FN mm_dd mm_log_dd(double x) {
  uint64_t ix = mm_bits(x);
  int64_t kadj = 0;
  if (ix < 0x0010000000000000ull) { /* subnormal */
    ix = mm_bits(x * 4503599627370496.0);
    kadj = -52;
  }
  const uint64_t tmp = ix - 0x3FE6A09E667F3BCDull;
  const int64_t k = ((int64_t)tmp >> 52) + kadj;
  const unsigned i = (unsigned)((tmp >> 45) & (MM_LOG_N - 1));
  const double z = mm_from_bits(ix - (tmp & 0xFFF0000000000000ull));
  const double c = MM_LOG_C[i];
  const double d = z - c; /* exact */
  /* r = d / c as a double-double */
  const double rh = d / c;
  const mm_dd p = mm_two_prod(rh, c);
  const double rl = ((d - p.hi) - p.lo) / c;
  /* log1p(r) = r - r^2/2 + r^3 (1/3 - r/4 + r^2/5 - r^3/6 + r^4/7 - r^5/8 + r^6/9), |r| < 2^-7 */
  mm_dd sq = mm_two_prod(rh, rh);
  sq.lo = sq.lo + 2.0 * rh * rl;
  const double t =
      rh * sq.hi *
      (0.33333333333333331 +
       rh * (-0.25 + rh * (0.2 + rh * (-0.16666666666666666 +
                                       rh * (0.14285714285714285 + rh * (-0.125 + rh * 0.1111111111111111))))));
  const double kd = (double)k;
  const mm_dd s1 = mm_two_sum(kd * MM_LN2_HI, MM_LOG_LOGC_HI[i]);
  const mm_dd s2 = mm_two_sum(s1.hi, rh);
  const mm_dd s3 = mm_two_sum(s2.hi, -0.5 * sq.hi);
  const double lo =
      ((((((s1.lo + s2.lo) + s3.lo) + kd * MM_LN2_LO) + MM_LOG_LOGC_LO[i]) + rl) - 0.5 * sq.lo) + t;
  return mm_fast_two_sum(s3.hi, lo);
}

/* exp(h + l), |l| << |h| */
MM_FN double mm_exp_dd(double h, double l) {
  if (!(h == h)) return h;
  if (h > 709.782712893384) return MM_INF;
  if (h < -745.1332191019412) return 0.0;
  const double z = h * MM_INV_LN2_N;
  const double kd = (z + 6755399441055744.0) - 6755399441055744.0; /* round to nearest integer */
  const int64_t k = (int64_t)kd;
  const double r = ((h - kd * MM_LN2_HI_N) - kd * MM_LN2_LO_N) + l;
  const unsigned j = (unsigned)((uint64_t)k & (MM_EXP_N - 1));
  const int64_t e = (k - (int64_t)j) / MM_EXP_N;
  const double th = MM_EXP_T_HI[j], tl = MM_EXP_T_LO[j];
  const double r2 = r * r;
  const double p =
      r + r2 * (0.5 + r * (0.16666666666666666 +
                           r * (0.041666666666666664 + r * (0.0083333333333333332 + r * 0.0013888888888888889))));
  const double res = th + (tl + th * p); /* in [1, 2 + eps) */
  if (e >= -1021 && e <= 1022) return res * mm_from_bits((uint64_t)(1023 + e) << 52);
  if (e > 1022) return (res * mm_from_bits((uint64_t)(1023 + e - 600) << 52)) * mm_from_bits((uint64_t)(1023 + 600) << 52);
  return (res * mm_from_bits((uint64_t)(1023 + e + 600) << 52)) * mm_from_bits((uint64_t)(1023 - 600) << 52);
}

MM_FN double mm_log(double x) {
  const uint64_t ix = mm_bits(x);
  if (ix == 0x3FF0000000000000ull) return 0.0;
  if (ix - 1 >= 0x7FF0000000000000ull - 1) { /* 0, negative, inf, nan */
    if ((ix << 1) == 0) return -MM_INF;
    if (ix == 0x7FF0000000000000ull) return x;
    if ((ix >> 63) != 0 && (ix << 1) <= 0xFFE0000000000000ull) return MM_NAN; /* negative */
    return x + x;                                                                /* nan */
  }
  return mm_log_dd(x).hi;
}

MM_FN double mm_exp(double x) { return mm_exp_dd(x, 0.0); }

/* y is an integer: 0 no, 1 odd, 2 even */
MM_FN int mm_int_kind(double y) {
  const uint64_t iy = mm_bits(y) & 0x7FFFFFFFFFFFFFFFull;
  const int e = (int)(iy >> 52) - 1023;
  if (e < 0) return 0;
  if (e > 52) return 2;
  const uint64_t frac_mask = (e == 52) ? 0 : ((1ull << (52 - e)) - 1);
  if ((iy & frac_mask) != 0) return 0;
  return ((iy >> (52 - e)) & 1) ? 1 : 2;
}

MM_FN double mm_pow(double x, double y) {
  const uint64_t ix = mm_bits(x), iy = mm_bits(y);
  if ((iy << 1) == 0) return 1.0;              /* x^0 */
  if (ix == 0x3FF0000000000000ull) return 1.0; /* 1^y */
  if (!(x == x) || !(y == y)) return x + y;
  const double ax = mm_from_bits(ix & 0x7FFFFFFFFFFFFFFFull);
  const int neg = (int)(ix >> 63);
  const int kind = neg ? mm_int_kind(y) : 2;
  if (neg && ax != 0.0 && kind == 0 && ax != MM_INF) return MM_NAN;
  const double sign = (neg && kind == 1) ? -1.0 : 1.0;
  if (ax == 0.0) return (iy >> 63) ? sign * MM_INF : sign * 0.0;
  if ((iy & 0x7FFFFFFFFFFFFFFFull) == 0x7FF0000000000000ull) { /* y = +-inf */
    if (ax == 1.0) return 1.0;
    return ((ax > 1.0) == ((iy >> 63) == 0)) ? MM_INF : 0.0;
  }
  if (ax == MM_INF) return (iy >> 63) ? sign * 0.0 : sign * MM_INF;
  const mm_dd lg = mm_log_dd(ax);
  const mm_dd e = mm_two_prod(y, lg.hi);
  if (!(e.hi == e.hi) || e.hi > 1e300 || e.hi < -1e300) return e.hi > 0 ? sign * MM_INF : sign * 0.0;
  return sign * mm_exp_dd(e.hi, e.lo + y * lg.lo);
}

#endif

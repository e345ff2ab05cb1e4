// fold_model.cpp -- TEST INFRASTRUCTURE: a scalar rendition, lane by lane, of the wave-parallel
// left-to-right fp64 fold of modle_amd/csrc/sim_burnin.h (fold_terms_exact), checked here against
// the plain sequential sum std::accumulate performs (reference: src/stats/descriptive_impl.hpp:52-63,
// sum_of_squared_deviations) on hundreds of millions of terms of every shape that matters: ties of the
// rounding, running sums that cross a binade inside a batch, zero and huge terms, tiny running sums.
//
// The algorithm: while the running sum s stays inside one binade [B, 2B), with u = ulp(s),
//   fl(s + t) = s + q(t),  q(t) = t rounded to the nearest multiple of u,
// unless the rounding is a tie (then the parity of s / u decides) or the sum reaches 2B.  q(t) is
// (B + t) - B in fp64 arithmetic; sums of multiples of u below 2B are exact in any order, so a batch
// of 64 terms folds with ONE prefix sum -- up to the first lane that ties or crosses, which takes a
// true addition, and the rest of the batch starts again from there.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

static inline uint64_t bits_of(double x) { uint64_t b; std::memcpy(&b, &x, 8); return b; }
static inline double from_bits(uint64_t b) { double x; std::memcpy(&x, &b, 8); return x; }

static double fold64_model(double s, const double* t, long* true_adds) {
  int start = 0;
  for (int it = 0; it < 3; ++it) {
    const uint64_t ef = (bits_of(s) >> 52) & 0x7FF;
    if (ef <= 53 || ef == 0x7FF) break;  // zero, tiny or not finite: the chain
    const double B = from_bits(ef << 52), top = B + B, u_half = from_bits((ef - 53) << 52);
    double q[64], P[64];
    bool bad[64];
    for (int k = 0; k < 64; ++k) q[k] = k < start ? 0.0 : (B + t[k]) - B;
    // inclusive prefix sum, in the order of a six-step lane scan (any order is exact where it matters)
    for (int k = 0; k < 64; ++k) P[k] = q[k];
    for (int d = 1; d < 64; d <<= 1) {
      double nxt[64];
      for (int k = 0; k < 64; ++k) nxt[k] = k >= d ? P[k] + P[k - d] : P[k];
      std::memcpy(P, nxt, sizeof(P));
    }
    int fb = 64;
    for (int k = 63; k >= start; --k) {
      bad[k] = std::fabs(t[k] - q[k]) == u_half || !(s + P[k] < top);
      if (bad[k]) fb = k;
    }
    if (fb == 64) return s + P[63];
    if (fb > start) s = s + P[fb - 1];
    s = s + t[fb];
    ++*true_adds;
    start = fb + 1;
    if (start == 64) return s;
  }
  for (int k = start; k < 64; ++k) s = s + t[k];  // the chain of dependent additions
  *true_adds += 64 - start;
  return s;
}

int main(int argc, char** argv) {
  const long rounds = argc > 1 ? std::atol(argv[1]) : 200000;
  std::mt19937_64 rng(12345);
  long fails = 0, true_adds = 0, batches = 0;
  for (long r = 0; r < rounds && fails < 10; ++r) {
    // one "cell": n loop sizes of a random shape, mean, then the terms (x - mean)^2
    const int shape = argc > 3 ? std::atoi(argv[3]) : static_cast<int>(rng() % 8);
    const int n = 1 + static_cast<int>(rng() % (argc > 2 ? std::atol(argv[2]) : 700));
    static double x[20064], t[20064 + 64];
    double total = 0;
    for (int i = 0; i < n; ++i) {
      uint64_t v;
      switch (shape) {
        case 0: v = rng() % 3000000; break;                    // loop sizes like the simulation's
        case 1: v = rng() % 4; break;                          // tiny integers: exact sums, many ties
        case 2: v = (rng() % 2) * 1000000; break;              // two values
        case 3: v = rng() % 2 ? 0 : rng() % 200000000; break;  // half the LEFs unbound
        case 4: v = 1ull << (rng() % 31); break;               // powers of two
        case 5: v = 12345; break;                              // all equal (terms ~ 0 or tiny)
        case 6: v = rng() % 64 == 0 ? 4000000000ull : rng() % 100; break;  // rare huge terms
        default: v = rng() % (1ull << (1 + rng() % 32)); break;
      }
      x[i] = static_cast<double>(v);
      total += x[i];
    }
    const double avg = shape == 1 && (rng() & 1) ? std::ldexp(std::floor(total), -10) : total / n;
    for (int i = 0; i < n; ++i) {
      const double d = x[i] - avg;
      t[i] = d * d;
    }
    for (int i = n; i < n + 64; ++i) t[i] = 0.0;  // (lanes past the end hold +0.0)
    double seq = 0.0, par = 0.0;
    for (int i = 0; i < n; ++i) seq = seq + t[i];
    for (int base = 0; base < n; base += 64) {
      par = fold64_model(par, t + base, &true_adds);
      ++batches;
    }
    if (bits_of(seq) != bits_of(par)) {
      std::printf("MISMATCH round %ld shape %d n %d: %.17g vs %.17g\n", r, shape, n, seq, par);
      ++fails;
    }
  }
  std::printf("%ld rounds, %ld batches, %.2f true additions per batch, %ld mismatches\n", rounds, batches,
              batches ? static_cast<double>(true_adds) / batches : 0.0, fails);
  return fails != 0;
}

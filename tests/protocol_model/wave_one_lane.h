// wave_one_lane.h -- a ONE-LANE rendition of the wavefront vocabulary of modle_amd/csrc/wave_hip.h.
//
// TEST INFRASTRUCTURE (tests/protocol_model): the hand-over protocol of the helper-wave mode
// (modle_amd/csrc/sim_pair.h, sim_helper.h, and the fed stream of sim_rng.h) is between WAVES, not
// between lanes, so here a wave is one host thread with one lane: collectives are identities, the
// words the waves hand work over with are accessed with exactly the operations the device code
// names (release stores, acquire loads, compare-and-swap, exchange), everything else the waves pass
// to each other is plain memory -- which is what lets ThreadSanitizer check the protocol: a plain
// access that the acquire / release pairs do not order is reported as a race.
// The lane emulator (tests/wave_emu) runs 64 lanes as fibers on ONE thread and cannot do this.
#pragma once
#include <math.h>
#include <sched.h>
#include <stdint.h>
#include <string.h>
#include <time.h>

#include "modle_math.h"

#define MODLE_DEV static inline __attribute__((always_inline))
#define MODLE_DEV_CALL static __attribute__((noinline))
#define MODLE_LDS
#define MODLE_DEV_MEMBER inline __attribute__((always_inline))
#define MODLE_DEV_NOINLINE static __attribute__((noinline))

namespace wave {

MODLE_DEV unsigned lane() { return 0; }
MODLE_DEV uint64_t ballot(bool p) { return p ? 1 : 0; }
MODLE_DEV bool any(bool p) { return p; }
template <class T>
MODLE_DEV T shfl(T v, unsigned) { return v; }
template <class T>
MODLE_DEV T bcast(T v, unsigned) { return v; }
template <class T>
MODLE_DEV T* as_global(T* p) { return p; }
template <class T>
MODLE_DEV T uniform(T v) { return v; }
template <class T>
MODLE_DEV T known_uniform(T v) { return v; }
template <class T>
MODLE_DEV T shfl_down(T v, unsigned) { return v; }
template <class T>
MODLE_DEV T shfl_up(T v, unsigned) { return v; }
MODLE_DEV uint32_t shfl_up1(uint32_t v) { return v; }
MODLE_DEV bool shfl_up1(bool v) { return v; }
enum ScanStep { SCAN_SHR1, SCAN_SHR2, SCAN_SHR4, SCAN_SHR8, SCAN_BCAST15, SCAN_BCAST31 };
template <int STEP>
MODLE_DEV uint32_t scan_move(uint32_t, uint32_t identity) { return identity; }

// One lane: program order is all the ordering a wave needs inside itself.  (Between waves only the
// named acquire / release operations below order anything: that is the point of the model.)
MODLE_DEV void sync_mem() {}
MODLE_DEV void sync_lds() {}
MODLE_DEV void lockstep() {}
MODLE_DEV uint64_t clock() { return 0; }
MODLE_DEV void pin(uint32_t&) {}
MODLE_DEV void launder(uint32_t&) {}
MODLE_DEV void sched_fence() {}
struct F64x2 {
  double v[2];
};
MODLE_DEV F64x2 lds_ld2_f64(const double* p) { return F64x2{{p[0], p[1]}}; }
MODLE_DEV double own_regs(double v) { return v; }
MODLE_DEV uint32_t own_regs(uint32_t v) { return v; }
MODLE_DEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
struct LdsRow {
  uint32_t h[8];
};
MODLE_DEV LdsRow lds_load_row(const uint64_t* table, uint32_t v) {
  LdsRow r;
  for (int i = 0; i < 4; ++i) {
    const uint64_t x = table[4 * v + i];
    r.h[2 * i] = static_cast<uint32_t>(x);
    r.h[2 * i + 1] = static_cast<uint32_t>(x >> 32);
  }
  return r;
}
template <class T>
MODLE_DEV T ld_stream(const T* p) { return *p; }
template <class T>
MODLE_DEV void st_stream(T* p, T v) { *p = v; }
struct U32x2 {
  uint32_t v[2];
};
MODLE_DEV U32x2 ld2(const uint32_t* p, uint32_t k) { return U32x2{{p[k], p[k + 1]}}; }
struct U32x4 {
  uint32_t v[4];
};
MODLE_DEV U32x4 ld4(const uint32_t* p, uint32_t k) { return U32x4{{p[k], p[k + 1], p[k + 2], p[k + 3]}}; }
MODLE_DEV U32x4 zero4() { return U32x4{{0, 0, 0, 0}}; }
MODLE_DEV void st4(uint32_t* p, uint32_t k, const U32x4& x) {
  for (int q = 0; q < 4; ++q) p[k + q] = x.v[q];
}
struct LdRaw {
  template <class T, class D>
  T operator()(const T* p, uint32_t k, bool ok, D dflt, T) const { return ok ? p[k] : static_cast<T>(dflt); }
};
struct LdMask {
  template <class T, class D>
  T operator()(const T*, uint32_t, bool ok, D dflt, T cur) const { return ok ? cur : static_cast<T>(dflt); }
};
template <class T, class D>
MODLE_DEV T ld_sel(const T* p, uint32_t k, bool ok, D dflt) { return ok ? p[k] : static_cast<T>(dflt); }

// ---- the operations of the hand-over protocol, as wave_hip.h names them ----
MODLE_DEV uint32_t load_system_u32(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
MODLE_DEV void st_release_wg(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
MODLE_DEV uint32_t ld_acquire_wg(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
MODLE_DEV bool cas_wg(uint32_t* p, uint32_t expected, uint32_t desired) {
  return __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED);
}
MODLE_DEV uint32_t exchange_wg(uint32_t* p, uint32_t v) { return __atomic_exchange_n(p, v, __ATOMIC_ACQ_REL); }
MODLE_DEV void nap() { sched_yield(); }
// the window of the "claim before first request" race, held open (regression build only)
MODLE_DEV void model_delay() {
  struct timespec ts = {0, 3 * 1000 * 1000};
  nanosleep(&ts, nullptr);
}
MODLE_DEV void atomic_inc_u32(uint32_t* p) { __atomic_fetch_add(p, 1u, __ATOMIC_RELAXED); }
MODLE_DEV void atomic_add_u64(uint64_t* p, uint64_t v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
MODLE_DEV uint32_t atomic_fetch_add_u32(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
MODLE_DEV uint32_t lds_fetch_add_u32(uint32_t* p, uint32_t v) {  // (lanes run one at a time)
  const uint32_t old = *p;
  *p = old + v;
  return old;
}
MODLE_DEV void lds_or_u32(uint32_t* p, uint32_t v) { *p |= v; }

MODLE_DEV double f_log(double x) { return mm_log(x); }
MODLE_DEV double f_exp(double x) { return mm_exp(x); }
MODLE_DEV double f_pow(double x, double y) { return mm_pow(x, y); }
MODLE_DEV double f_sqrt(double x) { return sqrt(x); }
MODLE_DEV double f_floor(double x) { return floor(x); }
MODLE_DEV double f_round(double x) { return round(x); }
MODLE_DEV double f_abs(double x) { return fabs(x); }
MODLE_DEV bool f_isfinite(double x) { return isfinite(x); }
MODLE_DEV int popc64(uint64_t x) { return __builtin_popcountll(x); }
MODLE_DEV int ctz64(uint64_t x) { return x ? __builtin_ctzll(x) : -1; }
MODLE_DEV int clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }

}  // namespace wave

// wave_emu.h -- CPU emulation of the wavefront vocabulary of modle_amd/csrc/wave_hip.h.
//
// TEST INFRASTRUCTURE.  The 64 lanes of a wave run as cooperative fibers on one host thread.  A
// lane runs until it reaches a collective (ballot / shuffle / sync), deposits its operand and
// yields to the next lane; when control comes back every lane has deposited, so the collective
// can be evaluated.  Slots are double-buffered because a lane can be at most one collective
// ahead of another.  Every collective carries the source line it was issued from and the
// emulator aborts when lanes meet at different lines: wave-divergent collectives (which would
// be undefined on hardware) are caught here.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "modle_math.h"  // the product's software log / exp / pow

#define MODLE_DEV static inline __attribute__((always_inline))
#define MODLE_DEV_CALL static __attribute__((noinline))
#define MODLE_LDS
#define MODLE_DEV_MEMBER inline __attribute__((always_inline))
#define MODLE_DEV_NOINLINE static __attribute__((noinline))

namespace wave_emu {

constexpr int kLanes = 64;

struct Slot {
  uint64_t v[2];
  int line;
};

struct WaveRuntime {
  void* lane_sp[kLanes];
  void* main_sp;
  char* stacks;
  int cur;
  bool done[kLanes];
  Slot slots[2][kLanes];
  unsigned coll_count[kLanes];
  void (*body)(void*);
  void* arg;
  // lane schedule: lanes run in the cyclic order order[0], order[1], ...; slot_of is the inverse.
  // On hardware all lanes execute an instruction together, so a correct kernel must not depend
  // on which lane runs first between two collectives: tests run it under several schedules.
  int order[kLanes];
  int slot_of[kLanes];
};

extern thread_local WaveRuntime* g_rt;
#ifdef MODLE_EMU_THREADS
// Race-detector build (make tsan_emu): every lane is a THREAD, the collectives are the only thing that
// orders them (a barrier each) -- so ThreadSanitizer reports every word of LDS or device memory that one
// lane writes and another touches without a collective in between: the barrier the GPU may not need
// (a wave's LDS operations execute in order) and the language does.
extern thread_local int t_lane;
void threads_barrier();
#endif

extern "C" void modle_emu_switch(void** save_sp, void* load_sp);

inline void yield_next() {
  WaveRuntime* rt = g_rt;
  const int me = rt->cur;
  int nxt = me;
  for (int k = 1; k <= kLanes; ++k) {
    const int c = rt->order[(rt->slot_of[me] + k) % kLanes];
    if (!rt->done[c]) {
      nxt = c;
      break;
    }
  }
  if (nxt == me) return;
  rt->cur = nxt;
  modle_emu_switch(&rt->lane_sp[me], rt->lane_sp[nxt]);
}

// deposit + rendezvous; returns the slot bank to read from
inline const Slot* collective(uint64_t lo, uint64_t hi, int line) {
  WaveRuntime* rt = g_rt;
#ifdef MODLE_EMU_THREADS
  {
    const int me = t_lane;
    const unsigned bank = rt->coll_count[me]++ & 1u;
    rt->slots[bank][me].v[0] = lo;
    rt->slots[bank][me].v[1] = hi;
    rt->slots[bank][me].line = line;
    threads_barrier();
    const Slot* s = rt->slots[bank];
    for (int l = 0; l < kLanes; ++l) {
      if (s[l].line != line) {
        fprintf(stderr, "wave_emu: divergent collective: lane %d at line %d, lane %d at line %d\n", me, line, l, s[l].line);
        abort();
      }
    }
    return s;
  }
#endif
  const int me = rt->cur;
  const unsigned bank = rt->coll_count[me]++ & 1u;
  rt->slots[bank][me].v[0] = lo;
  rt->slots[bank][me].v[1] = hi;
  rt->slots[bank][me].line = line;
  yield_next();
  const Slot* s = rt->slots[bank];
  for (int l = 0; l < kLanes; ++l) {
    if (s[l].line != line) {
      fprintf(stderr, "wave_emu: divergent collective: lane %d at line %d, lane %d at line %d\n",
              me, line, l, s[l].line);
      abort();
    }
  }
  return s;
}

// schedule: 0 = lanes 0..63, 1 = 63..0, otherwise the seed of a random permutation
void run_wave(void (*body)(void*), void* arg);
void set_lane_schedule(unsigned schedule);

}  // namespace wave_emu

namespace wave {

#ifdef MODLE_EMU_THREADS
MODLE_DEV unsigned lane() { return static_cast<unsigned>(wave_emu::t_lane); }
#else
MODLE_DEV unsigned lane() { return static_cast<unsigned>(wave_emu::g_rt->cur); }
#endif

MODLE_DEV uint64_t ballot(bool p, int line = __builtin_LINE()) {
  const wave_emu::Slot* s = wave_emu::collective(p ? 1 : 0, 0, line);
  uint64_t m = 0;
  for (int l = 0; l < 64; ++l) m |= (s[l].v[0] & 1ull) << l;
  return m;
}
MODLE_DEV bool any(bool p, int line = __builtin_LINE()) { return ballot(p, line) != 0; }

template <class T>
MODLE_DEV T shfl(T v, unsigned src, int line = __builtin_LINE()) {
  static_assert(sizeof(T) <= 16, "shuffle operand too wide");
  uint64_t w[2] = {0, 0};
  memcpy(w, &v, sizeof(T));
  const wave_emu::Slot* s = wave_emu::collective(w[0], w[1], line);
  T out;
  memcpy(&out, s[src & 63].v, sizeof(T));
  return out;
}
template <class T>
MODLE_DEV T bcast(T v, unsigned src, int line = __builtin_LINE()) {
  return shfl(v, src, line);
}
template <class T>
MODLE_DEV T* as_global(T* p) { return p; }

// the value must be identical in all lanes (checked)
template <class T>
MODLE_DEV T uniform(T v, int line = __builtin_LINE()) {
  static_assert(sizeof(T) <= 8, "uniform operand too wide");
  uint64_t w = 0;
  memcpy(&w, &v, sizeof(T));
  const wave_emu::Slot* s = wave_emu::collective(w, 0, line);
  for (int l = 1; l < 64; ++l) {
    if (s[l].v[0] != s[0].v[0]) {
      fprintf(stderr, "wave_emu: value declared uniform at line %d differs: lane 0 = %llx, lane %d = %llx\n",
              line, (unsigned long long)s[0].v[0], l, (unsigned long long)s[l].v[0]);
      abort();
    }
  }
  return v;
}
template <class T>
MODLE_DEV T shfl_down(T v, unsigned delta, int line = __builtin_LINE()) {
  const unsigned l = lane();
  const unsigned src = l + delta < 64 ? l + delta : l;
  return shfl(v, src, line);
}
template <class T>
MODLE_DEV T shfl_up(T v, unsigned delta, int line = __builtin_LINE()) {
  const unsigned l = lane();
  const unsigned src = l >= delta ? l - delta : l;
  return shfl(v, src, line);
}

MODLE_DEV uint32_t shfl_up1(uint32_t v, int line = __builtin_LINE()) { return shfl_up(v, 1u, line); }
MODLE_DEV bool shfl_up1(bool v, int line = __builtin_LINE()) { return shfl_up(v, 1u, line); }

enum ScanStep { SCAN_SHR1, SCAN_SHR2, SCAN_SHR4, SCAN_SHR8, SCAN_BCAST15, SCAN_BCAST31 };
template <int STEP>
MODLE_DEV uint32_t scan_move(uint32_t v, uint32_t identity, int line = __builtin_LINE()) {
  const wave_emu::Slot* s = wave_emu::collective(v, 0, line * 8 + STEP);
  const int l = static_cast<int>(lane());
  const int row = l / 16, in_row = l % 16;
  int src = -1;
  if (STEP <= SCAN_SHR8) {
    const int n = 1 << STEP;
    if (in_row >= n) src = l - n;
  } else if (STEP == SCAN_BCAST15) {
    if (row == 1 || row == 3) src = row * 16 - 1;
  } else {
    if (row >= 2) src = 31;
  }
  return src >= 0 ? static_cast<uint32_t>(s[src].v[0]) : identity;
}

MODLE_DEV void sync_mem(int line = __builtin_LINE()) { (void)wave_emu::collective(0, 0, line); }
MODLE_DEV void sync_lds(int line = __builtin_LINE()) { (void)wave_emu::collective(0, 0, line); }
MODLE_DEV void lockstep(int line = __builtin_LINE()) { (void)wave_emu::collective(0, 0, line); }

#ifndef MODLE_EMU_WRITE_TRACE
MODLE_DEV uint64_t clock() { return 0; }
#else
// Write-trace build (tools/emu_write_trace.py): compiled with MODLE_PHASE_TIMERS, whose PHASE macro reads
// the clock before and after every phase and adds the difference to the phase's slot.  Here the "clock"
// counts bytes: the first reading of a pair snapshots the cell's workspace and returns 0, the second
// returns the number of bytes that differ from the snapshot -- so the slots end up holding the bytes of
// device memory each phase CHANGED (a lower bound of what it wrote: a value written again does not
// count).  All lanes arrive (two barriers), lane 0 does the work.
}  // namespace wave
namespace emu_wtrace {
uint64_t tick();
}
namespace wave {
MODLE_DEV uint64_t clock(int line = __builtin_LINE()) {
  (void)wave_emu::collective(0, 0, line);
  uint64_t w[2] = {0, 0};
  if (lane() == 0) w[0] = emu_wtrace::tick();
  const wave_emu::Slot* s = wave_emu::collective(w[0], w[1], line);
  return s[0].v[0];
}
#endif
MODLE_DEV void pin(uint32_t&) {}
MODLE_DEV void launder(uint32_t&) {}
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
MODLE_DEV void sched_fence() {}
struct F64x2 {
  double v[2];
};
MODLE_DEV F64x2 lds_ld2_f64(const double* p) { return F64x2{{p[0], p[1]}}; }

template <class T>
MODLE_DEV T ld_stream(const T* p) { return *p; }
template <class T, class V>
MODLE_DEV void st_stream(T* p, V v) { *p = static_cast<T>(v); }
template <class T>
MODLE_DEV T known_uniform(T v) { return v; }
// four consecutive words as one 128-bit access: p + k must be 16-byte aligned (k a multiple of 4
// in an array that starts on a 16-byte boundary)
struct U32x2 {
  uint32_t v[2];
};
MODLE_DEV U32x2 ld2(const uint32_t* p, uint32_t k) {
  U32x2 x;
  x.v[0] = p[k];
  x.v[1] = p[k + 1];
  return x;
}
struct U32x4 {
  uint32_t v[4];
};
#ifdef MODLE_EMU_THREADS
// The device code's idiom for a lane that has nothing to load is "load element 0 and do not look at it"
// (no branch around the load).  Lane 0 may be storing element 0 meanwhile: a race by the letter, the one
// the detector is not after.  Loads of element 0 by any other lane are kept from it.
extern "C" void AnnotateIgnoreReadsBegin(const char* file, int line);
extern "C" void AnnotateIgnoreReadsEnd(const char* file, int line);
struct DummyLoadGuard {
  bool on;
  explicit DummyLoadGuard(uint32_t k) : on(k == 0 && wave_emu::t_lane != 0) {
    if (on) AnnotateIgnoreReadsBegin(__FILE__, __LINE__);
  }
  ~DummyLoadGuard() {
    if (on) AnnotateIgnoreReadsEnd(__FILE__, __LINE__);
  }
};
#define MODLE_EMU_DUMMY_LOAD(k) DummyLoadGuard dummy_load_guard_(k)
#else
#define MODLE_EMU_DUMMY_LOAD(k)
#endif
MODLE_DEV U32x4 ld4(const uint32_t* p, uint32_t k) {
  MODLE_EMU_DUMMY_LOAD(k);
  U32x4 x;
  for (int q = 0; q < 4; ++q) x.v[q] = p[k + q];
  return x;
}
MODLE_DEV U32x4 ld4(const uint16_t* p, uint32_t k) {
  MODLE_EMU_DUMMY_LOAD(k);
  U32x4 x;
  for (int q = 0; q < 4; ++q) x.v[q] = p[k + q];
  return x;
}
MODLE_DEV void st4(uint16_t* p, uint32_t k, const U32x4& x) {
  for (int q = 0; q < 4; ++q) p[k + q] = static_cast<uint16_t>(x.v[q]);
}
MODLE_DEV U32x2 ld2(const uint16_t* p, uint32_t k) {
  U32x2 x;
  x.v[0] = p[k];
  x.v[1] = p[k + 1];
  return x;
}
MODLE_DEV U32x4 zero4() { return U32x4{{0, 0, 0, 0}}; }
MODLE_DEV void st4(uint32_t* p, uint32_t k, const U32x4& x) {
  for (int q = 0; q < 4; ++q) p[k + q] = x.v[q];
}
// The two halves of ld_sel for loads that are requested one group ahead of their use: LdRaw at
// the request (no select, hence no wait, behind the load), LdMask where the values are consumed.
// A loader written as `r.x = op(p, k, ok, dflt, r.x)` serves both.
struct LdRaw {
  template <class T, class D, class R>
  R operator()(const T* p, uint32_t k, bool ok, D dflt, R) const {
    (void)dflt;
    MODLE_EMU_DUMMY_LOAD(k);
    return ok ? static_cast<R>(p[k]) : static_cast<R>(dflt);
  }
};
struct LdMask {
  template <class T, class D, class R>
  R operator()(const T*, uint32_t, bool ok, D dflt, R cur) const {
    return ok ? cur : static_cast<R>(dflt);
  }
};
template <class T, class D>
MODLE_DEV T ld_sel(const T* p, uint32_t k, bool ok, D dflt) {
  return ok ? p[k] : static_cast<T>(dflt);
}
MODLE_DEV uint32_t load_system_u32(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
// (hand-over words of the helper-wave mode: the emulator runs one wave, the mode is never on)
MODLE_DEV void st_release_wg(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
MODLE_DEV uint32_t ld_acquire_wg(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
MODLE_DEV bool cas_wg(uint32_t* p, uint32_t expected, uint32_t desired) {
  return __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED);
}
MODLE_DEV uint32_t exchange_wg(uint32_t* p, uint32_t v) { return __atomic_exchange_n(p, v, __ATOMIC_ACQ_REL); }
MODLE_DEV void nap() {}
MODLE_DEV double own_regs(double v) { return v; }
MODLE_DEV uint32_t own_regs(uint32_t v) { return v; }
MODLE_DEV void atomic_inc_u32(uint32_t* p) { __atomic_fetch_add(p, 1u, __ATOMIC_RELAXED); }
MODLE_DEV void atomic_add_u64(uint64_t* p, uint64_t v) {
  __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
MODLE_DEV uint32_t atomic_fetch_add_u32(uint32_t* p, uint32_t v) {
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
#ifdef MODLE_EMU_THREADS
MODLE_DEV uint32_t lds_fetch_add_u32(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
#else
MODLE_DEV uint32_t lds_fetch_add_u32(uint32_t* p, uint32_t v) {  // (lanes run one at a time)
  const uint32_t old = *p;
  *p = old + v;
  return old;
}
#endif
#ifdef MODLE_EMU_THREADS
MODLE_DEV void lds_or_u32(uint32_t* p, uint32_t v) { __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
#else
MODLE_DEV void lds_or_u32(uint32_t* p, uint32_t v) { *p |= v; }  // (lanes run one at a time)
#endif

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

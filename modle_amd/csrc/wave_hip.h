// wave_hip.h -- wavefront primitives used by sim_device.h, gfx950 (wave64) implementation.
//
// The device code talks to the hardware only through this small vocabulary (lane id, ballot,
// shuffle, broadcast, intra-wave memory ordering, atomics, fp64 math).  tests/wave_emu/ holds a
// second implementation of the same vocabulary that runs the 64 lanes as cooperative fibers on a
// CPU, so that the kernel logic can be stepped against the oracle without a GPU (test
// infrastructure; never shipped, never a fallback).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MODLE_DEV __device__ __forceinline__
#define MODLE_DEV_NOINLINE __device__ __noinline__

namespace wave {

MODLE_DEV unsigned lane() { return __lane_id(); }
MODLE_DEV uint64_t ballot(bool p) { return __ballot(p); }
MODLE_DEV bool any(bool p) { return __ballot(p) != 0; }

template <class T>
MODLE_DEV T shfl(T v, unsigned src) {
  return __shfl(v, static_cast<int>(src), 64);
}
// value held by lane `src` (src must be wave-uniform)
template <class T>
MODLE_DEV T bcast(T v, unsigned src) {
  return __shfl(v, static_cast<int>(src), 64);
}
// lane l receives the value of lane l+delta (own value when out of range)
template <class T>
MODLE_DEV T shfl_down(T v, unsigned delta) {
  return __shfl_down(v, delta, 64);
}
template <class T>
MODLE_DEV T shfl_up(T v, unsigned delta) {
  return __shfl_up(v, delta, 64);
}

// Orders this wave's earlier global/LDS stores before later loads issued by any lane of the wave.
MODLE_DEV void sync_mem() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
// Lanes run in lockstep on hardware: only a compiler-level scheduling barrier is needed where one
// lane overwrites data another lane has just read.  (The emulator yields here.)
MODLE_DEV void lockstep() { __builtin_amdgcn_wave_barrier(); }

MODLE_DEV void atomic_inc_u32(uint32_t* p) { atomicAdd(p, 1u); }
MODLE_DEV void atomic_add_u64(uint64_t* p, uint64_t v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(v));
}
MODLE_DEV uint32_t atomic_fetch_add_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }

MODLE_DEV double f_log(double x) { return ::log(x); }
MODLE_DEV double f_exp(double x) { return ::exp(x); }
MODLE_DEV double f_pow(double x, double y) { return ::pow(x, y); }
MODLE_DEV double f_sqrt(double x) { return ::sqrt(x); }
MODLE_DEV double f_floor(double x) { return ::floor(x); }
MODLE_DEV double f_round(double x) { return ::round(x); }
MODLE_DEV double f_abs(double x) { return ::fabs(x); }
MODLE_DEV bool f_isfinite(double x) { return ::isfinite(x); }

MODLE_DEV int popc64(uint64_t x) { return __popcll(x); }
MODLE_DEV int ctz64(uint64_t x) { return __ffsll(static_cast<unsigned long long>(x)) - 1; }
MODLE_DEV int clz64(uint64_t x) { return __clzll(static_cast<long long>(x)); }

}  // namespace wave

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

// software log / exp / pow shared with the CPU oracle (parity by construction, modle_math.h)
#define MM_FN __device__ inline
#define MM_TABLE __device__ static const
#include "modle_math.h"

#define MODLE_DEV __device__ __forceinline__
// a real call even in the default build (large, rarely changing helpers: keeps the kernel small)
#define MODLE_DEV_CALL __device__ __noinline__
// pointer into LDS with an explicit address space (needed across a real call, where the
// compiler cannot infer it)
#define MODLE_LDS __attribute__((address_space(3)))
#define MODLE_DEV_MEMBER __device__ __forceinline__
// Phase functions are inlined into the kernel by default: a wave-uniform value stays in scalar
// registers across phases and wave-uniform branches compile to scalar branches.  MODLE_OUTLINE
// turns them into real calls (shorter builds while debugging).
#ifdef MODLE_OUTLINE
#define MODLE_DEV_NOINLINE __device__ __noinline__
#else
#define MODLE_DEV_NOINLINE __device__ __forceinline__
#endif

namespace wave {

// The lane id is laundered through an empty asm statement: every phase gets a fresh opaque
// value, so the optimizer cannot hoist the per-lane address arithmetic of all phases out of the
// epoch loop (which kept hundreds of registers live and spilled them to scratch).
MODLE_DEV unsigned lane() {
  unsigned l = __lane_id();
  asm volatile("" : "+v"(l));
  return l;
}
MODLE_DEV uint64_t ballot(bool p) { return __ballot(p); }
MODLE_DEV bool any(bool p) { return __ballot(p) != 0; }

template <class T>
MODLE_DEV T shfl(T v, unsigned src) {
  return __shfl(v, static_cast<int>(src), 64);
}
// Declares that a pointer (read from a struct or from memory, where the compiler only knows a
// generic address) points to device memory: accesses through the result are global_* instead
// of flat_* instructions (scalar base + lane offset addressing, no coupling with LDS traffic).
template <class T>
MODLE_DEV T* as_global(T* p) {
  return (T*)((__attribute__((address_space(1))) T*)p);
}

// Declares a value that is identical in all lanes to the compiler (v_readfirstlane): it then lives
// in scalar registers and branches on it are scalar branches instead of exec-mask regions.  (The
// CPU lane emulator checks that the lanes really agree.)
MODLE_DEV uint32_t uniform(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
}
MODLE_DEV int32_t uniform(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
MODLE_DEV uint64_t uniform(uint64_t v) {
  const uint32_t lo = uniform(static_cast<uint32_t>(v));
  const uint32_t hi = uniform(static_cast<uint32_t>(v >> 32));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
MODLE_DEV int64_t uniform(int64_t v) { return static_cast<int64_t>(uniform(static_cast<uint64_t>(v))); }
MODLE_DEV bool uniform(bool v) { return uniform(static_cast<uint32_t>(v)) != 0; }
MODLE_DEV double uniform(double v) {
  return __longlong_as_double(static_cast<long long>(uniform(static_cast<uint64_t>(__double_as_longlong(v)))));
}
template <class T>
MODLE_DEV T* uniform(T* p) {
  return reinterpret_cast<T*>(uniform(reinterpret_cast<uint64_t>(p)));
}

// value held by lane `src` (src must be wave-uniform); the result is wave-uniform (v_readlane)
MODLE_DEV uint32_t bcast(uint32_t v, unsigned src) {
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>(src)));
}
MODLE_DEV int32_t bcast(int32_t v, unsigned src) {
  return __builtin_amdgcn_readlane(v, static_cast<int>(src));
}
MODLE_DEV uint64_t bcast(uint64_t v, unsigned src) {
  const uint32_t lo = bcast(static_cast<uint32_t>(v), src);
  const uint32_t hi = bcast(static_cast<uint32_t>(v >> 32), src);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
MODLE_DEV int64_t bcast(int64_t v, unsigned src) {
  return static_cast<int64_t>(bcast(static_cast<uint64_t>(v), src));
}
MODLE_DEV bool bcast(bool v, unsigned src) { return bcast(static_cast<uint32_t>(v), src) != 0; }
MODLE_DEV double bcast(double v, unsigned src) {
  return __longlong_as_double(static_cast<long long>(bcast(static_cast<uint64_t>(__double_as_longlong(v)), src)));
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

// lane l receives the value of lane l-1 (lane 0 keeps its own): one DPP lane move across the
// whole wave (wave_shr:1) instead of a ds_bpermute round trip
MODLE_DEV uint32_t shfl_up1(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x138, 0xF, 0xF, false));
}
MODLE_DEV bool shfl_up1(bool v) { return shfl_up1(static_cast<uint32_t>(v)) != 0; }

// One step of a 64-lane inclusive prefix scan made of DPP lane moves (no LDS round trip).  Lane
// l receives the value of the lane the step names, or `identity` when the step gives it none:
//   SCAN_SHR1/2/4/8  lane l - n of the same row of 16 lanes
//   SCAN_BCAST15     last lane of the previous row, rows 1 and 3 only
//   SCAN_BCAST31     lane 31, rows 2 and 3 only
// Applying  v = op(v, scan_move<S>(v, identity))  for the six steps in this order leaves in
// every lane the combination of lanes 0..l (op associative, identity neutral on the right).
enum ScanStep { SCAN_SHR1, SCAN_SHR2, SCAN_SHR4, SCAN_SHR8, SCAN_BCAST15, SCAN_BCAST31 };
template <int STEP>
MODLE_DEV uint32_t scan_move(uint32_t v, uint32_t identity) {
  constexpr int ctrl = STEP == SCAN_SHR1 ? 0x111 : STEP == SCAN_SHR2 ? 0x112 : STEP == SCAN_SHR4 ? 0x114
                     : STEP == SCAN_SHR8 ? 0x118 : STEP == SCAN_BCAST15 ? 0x142 : 0x143;
  constexpr int row_mask = STEP == SCAN_BCAST15 ? 0xA : STEP == SCAN_BCAST31 ? 0xC : 0xF;
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(identity), static_cast<int>(v),
                                                           ctrl, row_mask, 0xF, false));
}

// Orders this wave's earlier global/LDS stores before later loads issued by any lane of the wave.
// (Measured with the fence at wavefront scope -- no wait at all by LLVM's AMDGPU memory model, the
// operations of one wavefront being performed in program order: parity stays green and the kernel
// is 1 % SLOWER, the number of `s_waitcnt vmcnt(0)` in the kernel does not drop (614 vs 644): the
// drains at the phase boundaries come from register dependencies and loop headers, not from this
// fence.  -DMODLE_SYNC_MEM_SCOPE='"wavefront"' repeats the experiment.)
#ifndef MODLE_SYNC_MEM_SCOPE
#define MODLE_SYNC_MEM_SCOPE "workgroup"
#endif
MODLE_DEV void sync_mem() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, MODLE_SYNC_MEM_SCOPE);
  __builtin_amdgcn_wave_barrier();
}
// Same for data exchanged through LDS only.  LDS operations of one wave execute in order, so no
// wait is needed: this is a compiler-level ordering point (it does not wait for outstanding
// device-memory traffic the way sync_mem does).
MODLE_DEV void sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// Lanes run in lockstep on hardware: only a compiler-level scheduling barrier is needed where one
// lane overwrites data another lane has just read.  (The emulator yields here.)
MODLE_DEV void lockstep() { __builtin_amdgcn_wave_barrier(); }

// an opaque copy of a lane-dependent value (see simulate_tasks in modle_hip.hip: keeps jump threading
// from routing lanes around a lane-0 block along a second back edge)
MODLE_DEV void launder(uint32_t& v) { asm volatile("" : "+v"(v)); }
// the value has to exist in a register at this point of the program: keeps the optimizer from
// sinking the computation that produces it (sched_fence only binds the instruction scheduler)
MODLE_DEV void pin(uint32_t& v) { asm volatile("" : "+v"(v)); }
// A wave-uniform value in scalar registers OF ITS OWN.  Fields of a struct that reaches the kernel
// as an argument are loaded sixteen registers at a time, and a field used inside a loop keeps the
// whole group alive: when the group is spilled, every iteration reloads all sixteen registers
// (v_readlane each) for the two it needs.
MODLE_DEV double own_regs(double v) {
  v = uniform(v);
  asm volatile("" : "+s"(v));
  return v;
}
MODLE_DEV uint32_t own_regs(uint32_t v) {
  v = uniform(v);
  asm volatile("" : "+s"(v));
  return v;
}
// a ^ b ^ c in one instruction (v_bitop3_b32)
MODLE_DEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return static_cast<uint32_t>(__builtin_amdgcn_bitop3_b32(a, b, c, 0x96));
}
// Row `v` of a 16-row LDS table of 4 x 64-bit words (32 bytes per row, the table 16-byte aligned):
// two 128-bit reads.  h[2 i], h[2 i + 1] = low / high half of word i.
struct LdsRow {
  uint32_t h[8];
};
MODLE_DEV LdsRow lds_load_row(const MODLE_LDS uint64_t* table, uint32_t v) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const MODLE_LDS u32x4* row = reinterpret_cast<const MODLE_LDS u32x4*>(table) + 2 * v;
  const u32x4 a = row[0], b = row[1];
  return LdsRow{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}

// two consecutive doubles of an LDS buffer as one 128-bit read (p 16-byte aligned)
struct alignas(16) F64x2 {
  double v[2];
};
MODLE_DEV F64x2 lds_ld2_f64(const double* p) { return *reinterpret_cast<const F64x2*>(p); }

// the instruction scheduler may not move anything across this point
MODLE_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// constant-rate (100 MHz) timestamp, for the profiling build
MODLE_DEV uint64_t clock() { return wall_clock64(); }

// Streaming accesses of the rank-ordered sweeps (read once / written once per pass): with
// MODLE_NT they carry the non-temporal hint, so that they do not push the lines the scattered
// 4-byte stores keep coming back to out of L2.
#ifdef MODLE_NT
template <class T>
MODLE_DEV T ld_stream(const T* p) { return __builtin_nontemporal_load(p); }
template <class T, class V>
MODLE_DEV void st_stream(T* p, V v) { __builtin_nontemporal_store(static_cast<T>(v), p); }
#elif defined(MODLE_L1_BYPASS)
// experiment: the loads of the sweeps go to L2 directly (agent scope: the vector L1 does not look
// them up, hence cannot stall on a line that is still on its way)
template <class T>
MODLE_DEV T ld_stream(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T, class V>
MODLE_DEV void st_stream(T* p, V v) { *p = static_cast<T>(v); }
#else
template <class T>
MODLE_DEV T ld_stream(const T* p) { return *p; }
template <class T, class V>
MODLE_DEV void st_stream(T* p, V v) { *p = static_cast<T>(v); }
#endif
// a value that is the same in every lane by construction, where the compiler cannot see it: keeps
// it in scalar registers and branches on it scalar (no exchange in the emulator)
template <class T>
MODLE_DEV T known_uniform(T v) { return uniform(v); }
// p[k] where `ok`, `dflt` elsewhere, without a divergent branch around the load: lanes that are
// not `ok` read p[0] (p must point at one readable element at least) and drop it.  The compiler keeps
// an `ok ? p[k] : dflt` as s_and_saveexec / s_cbranch_execz around every single load.
// four consecutive words as one 128-bit access: p + k must be 16-byte aligned (k a multiple of 4
// in an array that starts on a 16-byte boundary)
// two consecutive words as one 64-bit access (p + k 8-byte aligned)
struct alignas(8) U32x2 {
  uint32_t v[2];
};
// (Element k of an array is addressed as base + a 32-BIT byte offset: written `p[k]` the byte offset
// is a 64-bit value -- k * sizeof(T) may exceed 2^32 for all the compiler knows -- and every access
// costs a 64-bit shift-and-add into a pair of address registers; with the 32-bit offset the
// instruction takes the scalar base and one offset register that all the arrays of a sweep share.
// Arrays of the workspace hold at most 2^24 elements of 8 bytes.)
template <class T>
MODLE_DEV const T* at(const T* p, uint32_t k) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + static_cast<uint32_t>(k * static_cast<uint32_t>(sizeof(T))));
}
template <class T>
MODLE_DEV T* at(T* p, uint32_t k) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(p) + static_cast<uint32_t>(k * static_cast<uint32_t>(sizeof(T))));
}
MODLE_DEV U32x2 ld2(const uint32_t* p, uint32_t k) { return *reinterpret_cast<const U32x2*>(at(p, k)); }
struct alignas(16) U32x4 {
  uint32_t v[4];
};
MODLE_DEV U32x4 ld4(const uint32_t* p, uint32_t k) { return *reinterpret_cast<const U32x4*>(at(p, k)); }
MODLE_DEV void st4(uint32_t* p, uint32_t k, const U32x4& x) { *reinterpret_cast<U32x4*>(at(p, k)) = x; }
// the same on arrays of 16-bit values (NARROW class: LEF ids, moves): four consecutive elements as one 64-bit
// access (p + k 8-byte aligned), widened to / narrowed from 32 bits in registers
MODLE_DEV U32x4 ld4(const uint16_t* p, uint32_t k) {
  const U32x2 x = *reinterpret_cast<const U32x2*>(at(p, k));
  return U32x4{{x.v[0] & 0xFFFFu, x.v[0] >> 16, x.v[1] & 0xFFFFu, x.v[1] >> 16}};
}
MODLE_DEV void st4(uint16_t* p, uint32_t k, const U32x4& x) {
  *reinterpret_cast<U32x2*>(at(p, k)) = U32x2{{x.v[0] | (x.v[1] << 16), x.v[2] | (x.v[3] << 16)}};
}
MODLE_DEV U32x2 ld2(const uint16_t* p, uint32_t k) {
  const uint32_t x = *reinterpret_cast<const uint32_t*>(at(p, k));
  return U32x2{{x & 0xFFFFu, x >> 16}};
}
// Four zeros made on the spot.  A literal {0, 0, 0, 0} is loop-invariant: the optimizer builds it once at
// the top of the kernel, holds four registers for it through every phase and, when registers run
// short, SPILLS it -- a 16-byte scratch reload in front of every store of zeros inside the sweeps
// (round 4: two builds 2 % slower for that reason alone).
MODLE_DEV U32x4 zero4() {
  uint32_t z = 0;
  asm volatile("" : "+v"(z));
  return U32x4{{z, z, z, z}};
}
// The two halves of ld_sel for loads that are requested one group ahead of their use: LdRaw at
// the request (no select, hence no wait, behind the load), LdMask where the values are consumed.
// A loader written as `r.x = op(p, k, ok, dflt, r.x)` serves both.
// (R: the type of the register the value lands in -- 32 bits also where the array holds 16-bit elements)
struct LdRaw {
  template <class T, class D, class R>
  MODLE_DEV R operator()(const T* p, uint32_t k, bool ok, D dflt, R) const {
    (void)dflt;
    return static_cast<R>(ld_stream(at(p, ok ? k : 0u)));
  }
};
struct LdMask {
  template <class T, class D, class R>
  MODLE_DEV R operator()(const T*, uint32_t, bool ok, D dflt, R cur) const {
    return ok ? cur : static_cast<R>(dflt);
  }
};
template <class T, class D>
MODLE_DEV T ld_sel(const T* p, uint32_t k, bool ok, D dflt) {
  const T v = ld_stream(at(p, ok ? k : 0u));
  return ok ? v : static_cast<T>(dflt);
}
// word in host memory written by the host while the kernel runs (the abort word)
MODLE_DEV uint32_t load_system_u32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Words in LDS through which two waves of a workgroup hand work to each other (helper wave,
// sim_pair.h): a release store publishes everything the wave has written before it -- device
// memory included -- to the waves of its workgroup, an acquire load that sees the stored value
// makes those writes visible to the reader.
MODLE_DEV void st_release_wg(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
MODLE_DEV uint32_t ld_acquire_wg(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// One lane's compare-and-swap / exchange on such a word (call from ONE lane; workgroup scope)
MODLE_DEV bool cas_wg(uint32_t* p, uint32_t expected, uint32_t desired) {
  return __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_WORKGROUP);
}
MODLE_DEV uint32_t exchange_wg(uint32_t* p, uint32_t v) {
  return __hip_atomic_exchange(p, v, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// gives the issue slots of the SIMD to the other waves for a few dozen cycles (spin loops)
MODLE_DEV void nap() { __builtin_amdgcn_s_sleep(2); }
MODLE_DEV void atomic_inc_u32(uint32_t* p) { atomicAdd(p, 1u); }
MODLE_DEV void atomic_add_u64(uint64_t* p, uint64_t v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(v));
}
MODLE_DEV uint32_t atomic_fetch_add_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
// returns *p and adds v to it, on a word in LDS that other lanes of the wave may be adding to as well
// (ds_add_rtn_u32: the adds of one wave instruction to one word are performed one after the other)
MODLE_DEV uint32_t lds_fetch_add_u32(uint32_t* p, uint32_t v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
// *p |= v on a word in LDS that other lanes of the wave may be updating too (ds_or_b32)
MODLE_DEV void lds_or_u32(uint32_t* p, uint32_t v) {
  __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// log / exp / pow are the shared software routines of modle_math.h (the same source the oracle
// compiles), as real calls: ~15 call sites, most of them in rarely taken branches
MODLE_DEV_CALL double f_log(double x) { return mm_log(x); }
MODLE_DEV_CALL double f_exp(double x) { return mm_exp(x); }
MODLE_DEV_CALL double f_pow(double x, double y) { return mm_pow(x, y); }
MODLE_DEV double f_sqrt(double x) { return ::sqrt(x); }
MODLE_DEV double f_floor(double x) { return ::floor(x); }
MODLE_DEV double f_round(double x) { return ::round(x); }
MODLE_DEV double f_abs(double x) { return ::fabs(x); }
MODLE_DEV bool f_isfinite(double x) { return ::isfinite(x); }

MODLE_DEV int popc64(uint64_t x) { return __popcll(x); }
MODLE_DEV int ctz64(uint64_t x) { return __ffsll(static_cast<unsigned long long>(x)) - 1; }
MODLE_DEV int clz64(uint64_t x) { return __clzll(static_cast<long long>(x)); }

}  // namespace wave

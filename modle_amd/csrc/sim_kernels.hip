// sim_kernels.hip -- the gfx950 kernels of the library, compiled once per size class (sim_types.h): NARROW
// (16-bit LEF ids and moves: every real chromosome) and, with -DMODLE_WIDE, WIDE (32-bit).  The host side
// (modle_hip.hip) picks the class of a launch (modle_hip_size_class) and calls the launchers at the end of
// this file.
//
// One wavefront simulates one (interval, cell) task (reference seam:
// Simulation::simulate_one_cell, src/libmodle/cpu/simulation.cpp:896-986, called from
// src/libmodle/cpu/scheduler_simulate.cpp:240).  The launch is persistent: one 512-thread
// workgroup per CU, its 8 waves pull tasks (largest chromosomes first) from a device-side
// counter, the way the reference's worker threads drain the task queue
// (scheduler_simulate.cpp:190-271).  LDS holds what every wave of the workgroup shares (the
// GF(2) jump table of the PRNG block generator and the ziggurat layer tables) plus each wave's
// ring of raw PRNG outputs; per-cell LEF / barrier state lives in a per-wave slice of a device
// workspace.
#include <hip/hip_runtime.h>

#include "wave_hip.h"
// clang-format off
#include "sim_device.h"
// clang-format on
#include "sim_launch.h"

using namespace modle_dev;
using namespace modle_launch;

#if MODLE_WAVES_PER_CU == 8
#ifdef MODLE_WIDE
#define MODLE_CLS(x) x##_wide
#else
#define MODLE_CLS(x) x##_narrow
#endif
#elif MODLE_WAVES_PER_CU == 12
#ifdef MODLE_WIDE
#define MODLE_CLS(x) x##_wide12
#else
#define MODLE_CLS(x) x##_narrow12
#endif
#else
#error "sim_kernels.hip is built for 8 or 12 waves per workgroup"
#endif

namespace {

__device__ __forceinline__ Workspace device_carve(char* base, u32 max_lefs, u32 max_barriers, u32 hist_len,
                                                  char* lds_base = nullptr) {
  // mirrors modle_host::carve_workspace
  const u64 Lp = (static_cast<u64>(max_lefs) + 63) & ~u64(63);
  u64 pw = 1;
  const u32 ml = max_lefs < 64 ? 64 : max_lefs;
  while (pw < ml) pw <<= 1;
  Workspace ws;
  char* p = wave::as_global(base);
  ws.sort_keys = reinterpret_cast<u64*>(p);
  p += pw * 8;
  ws.hist = reinterpret_cast<f64*>(p);
  p += 2 * static_cast<u64>(hist_len) * 8;
#if defined(MODLE_EXP_LDS_WS) && !defined(MODLE_EXP_LDS_WS_OFF)
  // (the unit arrays, the barrier states and the lists of stalling barriers: in this wave's slice of LDS)
  if (lds_base != nullptr) p = lds_base;
#else
  (void)lds_base;
#endif
  u32* q = reinterpret_cast<u32*>(p);
  ws.r_pos = q + 0 * Lp;
  ws.r_id = reinterpret_cast<lefid_t*>(q + 1 * Lp);
  ws.r_move = reinterpret_cast<move_t*>(q + 2 * Lp);
  ws.r_coll = q + 3 * Lp;
  ws.f_pos = q + 4 * Lp;
  ws.f_id = reinterpret_cast<lefid_t*>(q + 5 * Lp);
  ws.f_move = reinterpret_cast<move_t*>(q + 6 * Lp);
  ws.f_coll = q + 7 * Lp;
  ws.epoch = q + 8 * Lp;
  ws.r_rank = q + 9 * Lp;
  ws.f_rank = q + 10 * Lp;
  ws.stall = q + 11 * Lp;
  for (u32 k = 0; k < NUM_TMP; ++k) ws.tmp[k] = q + (12 + static_cast<u64>(k)) * Lp;
  for (u32 d = 0; d < 2; ++d) ws.by_id_pos[d] = q + (12 + NUM_TMP + static_cast<u64>(d)) * Lp;
  p += static_cast<u64>(NUM_STATE_ARRAYS) * Lp * 4;
  ws.bar_active = reinterpret_cast<u8*>(p);
  const u64 Bp = (static_cast<u64>(max_barriers) + 63) & ~u64(63);
  p += Bp;
  u32* hq = reinterpret_cast<u32*>(p);
  ws.hit_pos[0] = hq;
  ws.hit_pos[1] = hq + Bp;
  ws.hit_idx[0] = hq + 2 * Bp;
  ws.hit_idx[1] = hq + 3 * Bp;
  ws.capacity_lefs = max_lefs;
  ws.capacity_barriers = max_barriers;
  return ws;
}

struct BlockLds {
  alignas(16) u64 jump[JUMP_TABLE_WORDS];  // rows are read 128 bits at a time
  f64 zig[kZigWords];
  u64 ring[kLdsSlots][RNG_RING];
  u64 rng_state[kLdsSlots][RNG_STATE_WORDS];
  u64 rng_snap[kLdsSlots][8];
  u64 sort_keys[kLdsSlots][SORT_LDS_CAP];
  u32 stage[kLdsSlots][STAGE_CAP];
  u32 pairbox[kLdsSlots][PAIR_WORDS];  // helper-wave mode: hand-over words of main wave w (sim_pair.h)
#ifdef MODLE_EXP_LDS_WS
  alignas(16) char ws[kLdsSlots][MODLE_EXP_LDS_WS];
#endif
};

__device__ __forceinline__ WaveLds make_wave_lds(BlockLds& s, int wave_in_block) {
  WaveLds l;
  l.ring = s.ring[wave_in_block];
  l.rng_state = s.rng_state[wave_in_block];
  l.rng_snap = s.rng_snap[wave_in_block];
  l.abort_flag = nullptr;
  l.mbox = nullptr;
  l.pair_dynamic = false;
  l.jump_table = s.jump;
  l.zig_norm_x = s.zig;
  l.zig_norm_y = s.zig + 129;
  l.zig_exp_x = s.zig + 258;
  l.zig_exp_y = s.zig + 258 + 257;
  l.sort_lds = s.sort_keys[wave_in_block];
  l.stage = s.stage[wave_in_block];
  l.phase_ticks = nullptr;
  l.state_log = nullptr;
  l.state_log_cap = 0;
  l.trace = nullptr;
  l.trace_cap = 0;
  return l;
}

__device__ __forceinline__ void load_block_tables(BlockLds& s, const DeviceTables& t, int nthreads) {
  // (hand-over words of the helper-wave mode -- the first lane-state words of a producer wave double
  // as its own: sequence numbers start from zero on both sides, no main wave is running yet)
  for (u32 i = threadIdx.x; i < static_cast<u32>(kLdsSlots) * PAIR_WORDS; i += nthreads) {
    s.pairbox[i / PAIR_WORDS][i % PAIR_WORDS] = 0;
    reinterpret_cast<u32*>(s.rng_state[i / PAIR_WORDS])[i % PAIR_WORDS] = 0;
  }
  const u64* jump = wave::as_global(t.jump);
  const f64* zig = wave::as_global(t.zig);
  for (u32 i = threadIdx.x; i < JUMP_TABLE_WORDS; i += nthreads) s.jump[i] = jump[i];
  for (u32 i = threadIdx.x; i < static_cast<u32>(kZigWords); i += nthreads) s.zig[i] = zig[i];
  __syncthreads();
}

// The task loop of a main wave.
__device__ __forceinline__ void simulate_tasks(const SimArgs& a, const WaveLds& lds, u32 slot, int wave_in_block,
                                               char* lds_ws = nullptr) {
#ifdef MODLE_EXP_LDS_WS
  const Workspace ws = device_carve(a.workspace + static_cast<u64>(slot) * a.workspace_stride,
                                    a.max_lefs, a.max_barriers, a.params.hist_len, lds_ws);
#else
  const Workspace ws = device_carve(a.workspace + static_cast<u64>(slot) * a.workspace_stride,
                                    a.max_lefs, a.max_barriers, a.params.hist_len);
#endif
  if (lds.pair_dynamic) pair_open(lds.mbox);  // this main wave is running: an idle wave may become its helper
  u32 finished_interval = 0xFFFFFFFFu;  // interval of the task this wave has just completed
#ifdef MODLE_PHASE_TIMERS
  const u64 t_enter = wave::clock();
#endif
  for (;;) {
    // Pop one task.  Only lane 0 touches the counter, so this block branches on the lane id; the
    // wave barrier (a convergent operation the optimizer may not duplicate) and the laundered
    // lane id keep jump threading from routing the other 63 lanes around the pop along a second
    // back edge, which would split the wave for the convergent operations that follow.
    wave::lockstep();
    u32 leader = wave::lane();
    asm volatile("" : "+v"(leader));
    u32 t = 0;
    if (leader == 0) {
      // the previous task's outputs are complete (release fence at the end of the iteration):
      // tell the host, which may start reducing that interval's matrix once the count is zero
      if (finished_interval != 0xFFFFFFFFu)
        __hip_atomic_fetch_sub(a.interval_remaining + finished_interval, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
      t = atomicAdd(wave::as_global(a.task_counter), 1u);
    }
    finished_interval = 0xFFFFFFFFu;
    t = wave::bcast(t, 0);  // scalar from here on: the queue loop is a scalar loop
    if (t >= a.n_tasks) {  // every wave leaves once the queue is empty
#ifdef MODLE_PHASE_TIMERS
      if (a.phase_ticks != nullptr && wave::lane() == 0) {
        // how long this wave had work for: sum, longest, shortest (the launch lasts as long as the longest)
        const unsigned long long busy = wave::clock() - t_enter;
        unsigned long long* const pt = reinterpret_cast<unsigned long long*>(wave::as_global(a.phase_ticks));
        atomicAdd(pt + 16, busy);
        atomicMax(pt + 17, busy);
        atomicMin(pt + 18, busy);
      }
#endif
      if (lds.mbox == nullptr) break;
      if (!lds.pair_dynamic) {
        pair_dismiss(lds.mbox);
      } else {
        pair_close(lds.mbox);  // no more hand-overs; a helper that had claimed this wave is dismissed
      }
      break;
    }
    if (wave::uniform(wave::load_system_u32(wave::as_global(a.abort_flag))) != 0) {
      // cancelled: tasks that never started are reported as such (all lanes store the same word)
      CellResult none;
      __builtin_memset(&none, 0, sizeof(none));
      wave::as_global(a.results)[t] = none;
      wave::as_global(a.status)[t] = ERR_CANCELLED;
      finished_interval = wave::as_global(a.tasks)[t].interval;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      continue;
    }
    const Task task = wave::as_global(a.tasks)[t];
    CellResult res;
    WaveLds lds_t = lds;
    lds_t.phase_ticks = wave::as_global(a.phase_ticks);
    lds_t.abort_flag = wave::as_global(a.abort_flag);
    if (a.state_log != nullptr) {
      lds_t.state_log = wave::as_global(a.state_log) + static_cast<u64>(t) * a.state_log_cap * STATE_LOG_WORDS;
      lds_t.state_log_cap = a.state_log_cap;
    }
    if (t == 0 && a.trace != nullptr) {
      lds_t.trace = a.trace;
      lds_t.trace_cap = a.trace_cap;
    }
#ifdef MODLE_PHASE_TIMERS
    const u64 t_task = wave::clock();
#endif
    const u32 st = simulate_cell(a.params, wave::as_global(a.intervals)[task.interval], task, ws, lds_t, res);
#ifdef MODLE_PHASE_TIMERS
    if (a.phase_ticks != nullptr) {  // (per task: start, end, wave slot -- MODLE_PROF_TASK_TIMES writes them out)
      u64* const tt = wave::as_global(a.phase_ticks) + 20 + 3 * static_cast<u64>(t);
      tt[0] = t_task;
      tt[1] = wave::clock();
      tt[2] = slot;
    }
#endif
    // all lanes store the same words (no lane-dependent branch at the end of the loop body)
    wave::as_global(a.results)[t] = res;
    wave::as_global(a.status)[t] = st;
    finished_interval = task.interval;
    // contact increments (memory-side atomics) and the result words are performed before the
    // completion count of the interval drops
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  }
}

__global__ __launch_bounds__(kThreadsPerBlock) void MODLE_CLS(modle_simulate_cells)(SimArgs a) {
  __shared__ BlockLds s;
  load_block_tables(s, a.tables, kThreadsPerBlock);
  const int wave_in_block = wave::uniform(static_cast<int>(threadIdx.x / 64));
  const u32 slot = blockIdx.x * kWavesPerBlock + wave_in_block;
#ifdef MODLE_EXP_LDS_WS
  // (measurement build: waves 0 and MODLE_EXP_LDS_STRIDE only, no helpers)
  if (wave_in_block % MODLE_EXP_LDS_STRIDE != 0 || wave_in_block / MODLE_EXP_LDS_STRIDE >= kLdsSlots) return;
  {
    const int ls = wave_in_block / MODLE_EXP_LDS_STRIDE;
    WaveLds lds = make_wave_lds(s, ls);
    simulate_tasks(a, lds, slot, wave_in_block, s.ws[ls]);
  }
#else
  WaveLds lds = make_wave_lds(s, wave_in_block);
  // Roles (sim_pair.h).  Launches that leave wave slots empty (pair_mains != 0): fixed trios of
  // main wave / helper / PRNG producer.  Launches that fill the slots: every wave is a main wave
  // (one cell per wave) until the queue is empty, and then the helper of a main wave of its
  // workgroup that is still running.
#ifdef MODLE_NO_HELPERS  // (measurement: what the helper and producer loops cost the main path by being in the kernel)
  const bool fixed = false, dynamic = false;
#else
  const bool fixed = a.pair_mains != 0;
  const bool dynamic = !fixed && a.tail_helpers != 0;
#endif
  int serve_main = -1;  // >= 0: this wave is the helper of that main wave
  u32* feed = nullptr;
  if (fixed) {
    // with one or two main waves per workgroup there are waves to spare: wave 2 + m produces the
    // PRNG blocks for the helper of main wave m while that helper draws the moves (pair_feed)
#ifndef MODLE_RNG_PHILOX
    if (a.pair_mains <= 2 && wave_in_block >= 2 && wave_in_block < 4) {
      const int m = wave_in_block - 2;
      if (static_cast<u32>(m) >= a.pair_mains) return;
      const WaveLds lm = make_wave_lds(s, m);
      pair_feed(lm.ring, lm.jump_table, lm.rng_state, lm.rng_snap, reinterpret_cast<u32*>(s.rng_state[wave_in_block]),
                wave::as_global(a.abort_flag));
      return;
    }
#endif
    const int main_wave = wave_in_block < kWavesPerBlock / 2 ? wave_in_block : kWavesPerBlock - 1 - wave_in_block;
    if (static_cast<u32>(main_wave) >= a.pair_mains) return;
#ifndef MODLE_RNG_PHILOX
    if (a.pair_mains <= 2) feed = reinterpret_cast<u32*>(s.rng_state[2 + main_wave]);
#endif
    if (wave_in_block != main_wave) serve_main = main_wave;
  }
  if (serve_main < 0) {
    if (static_cast<u32>(wave_in_block) >= a.active_waves) return;
    if (fixed || dynamic) lds.mbox = s.pairbox[wave_in_block];
    lds.pair_dynamic = dynamic;
    simulate_tasks(a, lds, slot, wave_in_block);
    if (!dynamic) return;
  }
  for (;;) {
    u32* mbox = nullptr;
    u32 seen = 0;  // (fixed roles: the request counter starts from zero)
    if (dynamic) {
      // the queue is empty: claim a main wave of this workgroup that is still running without a
      // helper (sim_pair.h: pair_claim)
      serve_main = pair_claim(&s.pairbox[0][0], kWavesPerBlock, wave_in_block, seen);
      if (serve_main < 0) return;
      mbox = s.pairbox[serve_main];
    } else {
      mbox = s.pairbox[serve_main];
    }
    {
      // the helper: the main wave's generator, tables and workspace, its own staging and sort buffers
      // (every other field of the context zero: no list, no filter, no error)
      Cell c{};
      c.p = &a.params;
      c.lds = make_wave_lds(s, serve_main);
      c.lds.stage = lds.stage;
      c.lds.sort_lds = lds.sort_lds;
      c.lds.abort_flag = wave::as_global(a.abort_flag);
      c.ws = device_carve(a.workspace + static_cast<u64>(blockIdx.x * kWavesPerBlock + serve_main) * a.workspace_stride,
                          a.max_lefs, a.max_barriers, a.params.hist_len);
      c.g.ring = c.lds.ring;
      c.g.jump = c.lds.jump_table;
      c.g.state = c.lds.rng_state;
      c.g.snap = c.lds.rng_snap;
      pair_serve(c, wave::as_global(a.intervals), mbox, feed, seen, a.test_fault);
    }
    if (!dynamic) return;
  }
#endif
}

#if MODLE_WAVES_PER_CU == 8  // (the phase / unit hooks: one build per size class is enough)
__global__ __launch_bounds__(64) void MODLE_CLS(modle_test_phases)(PhaseArgs a) {
  __shared__ BlockLds s;
  load_block_tables(s, a.tables, 64);
  const WaveLds lds = make_wave_lds(s, 0);
  const Workspace ws = device_carve(a.workspace, a.n, a.max_barriers, 4);
  u64 raws = 0;
  TestImage img;
  img.rev_pos = a.image + 0 * a.n;
  img.fwd_pos = a.image + 1 * a.n;
  img.epoch = a.image + 2 * a.n;
  img.rev_rank = a.image + 3 * a.n;
  img.fwd_rank = a.image + 4 * a.n;
  img.rev_moves = a.image + 5 * a.n;
  img.fwd_moves = a.image + 6 * a.n;
  img.rev_coll = a.image + 7 * a.n;
  img.fwd_coll = a.image + 8 * a.n;
  const u32 st = run_test_phases(a.params, a.interval, ws, lds, img, a.mask, a.n, a.prng, raws);
  if (wave::lane() == 0) {
    *a.raws_out = raws;
    *a.status_out = st;
  }
}

__global__ __launch_bounds__(64) void MODLE_CLS(modle_test_units)(UnitArgs a) {
  __shared__ BlockLds s;
  load_block_tables(s, a.tables, 64);
  const WaveLds lds = make_wave_lds(s, 0);
  const Workspace ws = device_carve(a.workspace, a.n, 0, 4);
  const u32 st = run_test_units(a.params, a.interval, ws, lds, a.what, wave::as_global(a.in), a.n,
                                wave::as_global(a.out));
  if (wave::lane() == 0) *a.status_out = st;
}

#endif

}  // namespace

#if MODLE_WAVES_PER_CU == 8 && !defined(MODLE_WIDE)
namespace {
// The memory side of a launch without its arithmetic: every wave streams `reps` times through the first `hot` bytes of
// `n_arrays` arrays of its own workspace slot (128-bit loads, a store to every fourth array).  The host times it on a
// freshly allocated workspace to learn which of the placements the driver handed out (modle_hip.hip: place_workspace).
__global__ __launch_bounds__(768) void modle_probe_workspace(char* base, size_t slot_stride, u32 waves_per_block, size_t array_stride,
                                                             u32 n_arrays, u32 hot, u32 reps) {
  const u32 wave_in_block = threadIdx.x / 64, lane = threadIdx.x % 64;
  if (wave_in_block >= waves_per_block) return;
  char* slot = base + (static_cast<size_t>(blockIdx.x) * waves_per_block + wave_in_block) * slot_stride;
  u32 acc = 0;
  for (u32 r = 0; r < reps; ++r) {
    for (u32 k = 0; k < n_arrays; ++k) {
      uint4* arr = reinterpret_cast<uint4*>(slot + k * array_stride);
      for (u32 off = lane; off < hot / 16; off += 64) {
        uint4 v = arr[off];
        acc += v.x ^ v.y ^ v.z ^ v.w;
        if ((k & 3u) == (r & 3u)) {
          v.x += 1;
          arr[off] = v;
        }
      }
    }
  }
  if (acc == 0x12345678u) *reinterpret_cast<u32*>(slot) = acc;  // (keeps the loads; the workspace holds nothing yet)
}
}  // namespace
namespace modle_launch {
void probe_workspace(char* base, size_t slot_stride, u32 n_blocks, u32 waves_per_block, size_t array_stride, u32 n_arrays, u32 hot,
                     u32 reps, hipStream_t stream) {
  hipLaunchKernelGGL(modle_probe_workspace, dim3(n_blocks), dim3(64 * waves_per_block), 0, stream, base, slot_stride,
                     waves_per_block, array_stride, n_arrays, hot, reps);
}
}  // namespace modle_launch
#endif

namespace modle_launch {
void MODLE_CLS(simulate)(int grid, hipStream_t stream, const SimArgs& a) {
  hipLaunchKernelGGL(MODLE_CLS(modle_simulate_cells), dim3(grid), dim3(kThreadsPerBlock), 0, stream, a);
}
#if MODLE_WAVES_PER_CU == 8
void MODLE_CLS(test_phases)(const PhaseArgs& a) {
  hipLaunchKernelGGL(MODLE_CLS(modle_test_phases), dim3(1), dim3(64), 0, nullptr, a);
}
void MODLE_CLS(test_units)(const UnitArgs& a) {
  hipLaunchKernelGGL(MODLE_CLS(modle_test_units), dim3(1), dim3(64), 0, nullptr, a);
}
#endif
}  // namespace modle_launch

// sim_launch.h -- what the host side of the library (modle_hip.hip) and the two builds of the kernels
// (sim_kernels.hip: NARROW and, with -DMODLE_WIDE, WIDE -- sim_types.h "size classes") share: the kernel
// arguments and the functions that enqueue the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "sim_types.h"

namespace modle_launch {
using namespace modle_dev;


constexpr int kWavesPerBlock = MODLE_WAVES_PER_CU;
constexpr int kThreadsPerBlock = kWavesPerBlock * 64;
// Measurement build (round 5, profiles/r05a/lds_residency_ceiling.txt): MODLE_EXP_LDS_WS=<bytes> gives
// MODLE_EXP_LDS_WAVES (1 or 2) waves of every workgroup -- waves 0 and MODLE_EXP_LDS_STRIDE (4: the same
// SIMD, 1: two SIMDs) -- a slice of LDS that holds the unit arrays, the barrier states and the lists of
// stalling barriers of their cell; with MODLE_EXP_LDS_WS_OFF the same waves keep them in device memory.
#ifdef MODLE_EXP_LDS_WS
#ifndef MODLE_EXP_LDS_WAVES
#define MODLE_EXP_LDS_WAVES 2
#endif
#ifndef MODLE_EXP_LDS_STRIDE
#define MODLE_EXP_LDS_STRIDE 4
#endif
constexpr int kLdsSlots = MODLE_EXP_LDS_WAVES;
#else
constexpr int kLdsSlots = kWavesPerBlock;
#endif

struct DeviceTables {
  const u64* jump;   // JUMP_TABLE_WORDS
  const f64* zig;    // norm_x[129] norm_y[129] exp_x[257] exp_y[257]
};
constexpr int kZigWords = 129 + 129 + 257 + 257;

struct SimArgs {
  Params params;
  DeviceTables tables;
  const Interval* intervals;
  const Task* tasks;
  CellResult* results;
  u32* status;        // one word per task
  u32* task_counter;
  const u32* abort_flag;  // raised by modle_hip_cancel while the kernel runs
  u32* interval_remaining;  // host-visible: tasks of every interval still to finish in this launch
  u64* trace;  // diagnostic per-epoch trace of task 0 (MODLE_HIP_TRACE) or nullptr
  u32 trace_cap;
  u32 pad2_;
  u64* phase_ticks;  // profiling build only
  u64* state_log;    // MODLE_STATE_LOG build: n_tasks x state_log_cap records, or nullptr
  u32 state_log_cap;
  u32 pad3_;
  char* workspace;
  u64 workspace_stride;
  u32 n_tasks;
  u32 max_lefs;
  u32 max_barriers;
  u32 active_waves;  // waves of every workgroup that pull tasks (diagnostic: MODLE_HIP_ACTIVE_WAVES)
  // helper-wave mode (sim_pair.h), chosen by the host for launches with at most half as many tasks
  // as wave slots: waves 0 .. pair_mains-1 of a workgroup pull tasks, wave 7-m is the helper of
  // main wave m (waves are dealt to the four SIMDs in turn: with one or two main waves per
  // workgroup every wave of a pair has a SIMD of its own); 0 = off
  u32 pair_mains;
  // launches that fill the slots: a wave that finds the queue empty becomes the helper of a main
  // wave of its workgroup that is still running (sim_pair.h: PAIR_STATE); 0 = off
  u32 tail_helpers;
  // tests only (MODLE_HIP_TEST_FAULT): a fault injected into the hand-over protocol (sim_helper.h)
  u32 test_fault;
};

struct PhaseArgs {
  Params params;
  DeviceTables tables;
  Interval interval;
  char* workspace;
  u32* image;  // TestImage: nine arrays of n words
  u32 mask;
  u32 n;
  u64 prng[4];
  u64* raws_out;
  u32* status_out;
  u32 max_barriers;
};

struct UnitArgs {
  Params params;
  DeviceTables tables;
  Interval interval;
  char* workspace;
  const u64* in;
  u64* out;
  u32* status_out;
  u32 what;
  u32 n;
};

// One set per build of sim_kernels.hip: size class (narrow / wide) x waves per workgroup (8: 256 VGPRs, 512-output
// PRNG blocks, 512-key LDS buffers; 12: 168 VGPRs, 256-output blocks, 256-key buffers -- round 5: three waves per
// SIMD are worth 1.2 % on launches whose epochs re-insert few units; modle_hip.hip picks per launch).  The
// phase / unit test kernels exist in the 8-wave builds only.
void simulate_narrow(int grid, hipStream_t stream, const SimArgs& a);
void simulate_wide(int grid, hipStream_t stream, const SimArgs& a);
void simulate_narrow12(int grid, hipStream_t stream, const SimArgs& a);
void simulate_wide12(int grid, hipStream_t stream, const SimArgs& a);
void test_phases_narrow(const PhaseArgs& a);
void test_phases_wide(const PhaseArgs& a);
void test_units_narrow(const UnitArgs& a);
void test_units_wide(const UnitArgs& a);
// (sim_kernels.hip, NARROW 8-wave build: the streaming probe behind modle_hip.hip's place_workspace)
void probe_workspace(char* base, size_t slot_stride, u32 n_blocks, u32 waves_per_block, size_t array_stride, u32 n_arrays, u32 hot,
                     u32 reps, hipStream_t stream);

}  // namespace modle_launch

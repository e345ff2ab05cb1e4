// modle_hip.hip -- host half of the C ABI (include/modle_hip.h); the gfx950 kernels are in sim_kernels.hip.
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
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "modle_hip.h"
#include "wave_hip.h"
// clang-format off
#include "sim_device.h"  // (constants and plain types of the device code; the kernels live in sim_kernels.hip)
// clang-format on
#include "sim_launch.h"
#include "host_prng.hpp"
#include "launch_common.hpp"
#include "zig_tables.h"

using namespace modle_dev;

using namespace modle_launch;

namespace {

// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void set_err(char* err, size_t errlen, const std::string& msg) {
  if (err != nullptr && errlen != 0) std::snprintf(err, errlen, "%s", msg.c_str());
}

#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    const hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) {                                                               \
      set_err(err, errlen, std::string(#expr) + ": " + hipGetErrorString(e_));            \
      return MODLE_HIP_ERR_DEVICE;                                                        \
    }                                                                                     \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { reset(); }
  void reset() {
    if (p != nullptr) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  hipError_t ensure(size_t count) {
    if (count <= n) return hipSuccess;
    reset();
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  void swap(DevBuf& o) {
    std::swap(p, o.p);
    std::swap(n, o.n);
  }
};

struct IntervalRec {
  uint64_t start = 0, end = 0;
  uint64_t nrows = 0, ncols = 0;
  size_t n_barriers = 0;
  DevBuf<u32> bar_pos;
  DevBuf<u8> bar_dir;
  DevBuf<f64> bar_stp;  // stp_active | stp_inactive | occupancy
  DevBuf<u32> bar_bucket;
  size_t n_buckets = 0;
  DevBuf<u32> own_contacts;
  DevBuf<u64> own_occupancy;
  DevBuf<u64> missed;
  u32* d_contacts = nullptr;
  u64* d_occupancy = nullptr;
  std::vector<modle_hip_task> pending;
  std::vector<modle_hip_cell_result> results;  // submission order
  std::vector<size_t> launch_slots;            // index into the launch's task array
};

}  // namespace

struct modle_hip_handle {
  modle_hip_config cfg;
  Params params;
  int device = 0;
  int num_cus = 0;
  std::vector<std::unique_ptr<IntervalRec>> intervals;
  DevBuf<u64> d_jump;  // T^512 (8-wave kernels: blocks of 512 outputs; 12-wave kernels: pairs of blocks of 256)
  DevBuf<f64> d_zig;
  DevBuf<Interval> d_intervals;
  DevBuf<Task> d_tasks;
  DevBuf<CellResult> d_results;
  DevBuf<u32> d_status;
  DevBuf<u32> d_counter;
  // The abort word lives in host-mapped memory: modle_hip_cancel raises it with a plain store
  // while the persistent kernel holds every CU (anything that goes through a stream -- a fill
  // kernel, a stream memory operation, a small copy -- is executed by a kernel of its own and
  // waits for a free CU, i.e. for the launch to end), and the waves read it over the fabric once
  // every few epochs.
  u32* h_abort = nullptr;
  u32* d_abort = nullptr;  // device address of the same word
  u32* h_remaining = nullptr;  // host-mapped per-interval completion counters of the launch
  u32* d_remaining = nullptr;
  size_t remaining_cap = 0;
  // read / written by modle_hip_cancel and modle_hip_interval_done, which may run on another host
  // thread than the one that launches and waits
  std::atomic<bool> cancelled{false};
  // orders the reset of the abort word at a launch against modle_hip_cancel raising it
  std::mutex abort_mu;
  bool timing_valid = false;  // both events of the last launch were recorded
  void* trace_host = nullptr;           // MODLE_HIP_TRACE_SHM mapping, registered once per handle
  size_t trace_bytes = 0;
  u64* trace_dev = nullptr;
  DevBuf<char> d_workspace;
  uint64_t ws_tries = 0;        // candidates probed when the workspace was allocated (place_workspace)
  float ws_probe_ms = 0.0f;     // streaming probe on the one that was kept ...
  float ws_probe_worst_ms = 0.0f;  // ... and on the slowest candidate
  DevBuf<u64> d_phase_out;
  DevBuf<u64> d_trace;
  DevBuf<u64> d_phase_ticks;
  DevBuf<u64> d_state_log;
  u32 state_log_cap = 0;  // epochs logged per task (0 = off)
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  hipStream_t stream = nullptr;
  std::atomic<bool> in_flight{false};
  size_t n_launched = 0;
  std::vector<std::pair<int, size_t>> launch_map;  // launch task -> (interval, submission idx)
  float last_ms = 0.0f;
  // deadline of modle_hip_wait, counted from the launch (MODLE_HIP_WAIT_TIMEOUT_S /
  // modle_hip_set_wait_timeout), and how long an aborted launch gets to drain
  double wait_timeout_s = 0.0;  // 0: no deadline (the reference has none); bench.py and the tests set one
  double drain_timeout_s = 60.0;
  std::chrono::steady_clock::time_point launched_at;
  modle_hip_launch_info last_launch{};
};

extern "C" {

modle_hip_handle* modle_hip_create(const modle_hip_config* c, int device, char* err,
                                   size_t errlen) {
  if (c == nullptr) {
    set_err(err, errlen, "null config");
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_err(err, errlen, "no HIP device available: the MI355X path has no CPU fallback");
    return nullptr;
  }
  if (device < 0 || device >= ndev) {
    set_err(err, errlen, "invalid device ordinal");
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) {
    set_err(err, errlen, "hipSetDevice failed");
    return nullptr;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    set_err(err, errlen, "hipGetDeviceProperties failed");
    return nullptr;
  }
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
    set_err(err, errlen, std::string("unsupported GPU architecture ") + prop.gcnArchName +
                             " (this library is built for gfx950 only)");
    return nullptr;
  }
  auto h = std::make_unique<modle_hip_handle>();
  h->cfg = *c;
  h->params = modle_host::make_params(*c);
  h->device = device;
  h->num_cus = prop.multiProcessorCount;
  if (const char* e = std::getenv("MODLE_HIP_WAIT_TIMEOUT_S"); e != nullptr && std::atof(e) > 0.0)
    h->wait_timeout_s = std::atof(e);
  if (const char* e = std::getenv("MODLE_HIP_DRAIN_TIMEOUT_S"); e != nullptr && std::atof(e) > 0.0)
    h->drain_timeout_s = std::atof(e);
  static_assert(RNG_BLOCK == 512, "the host half is compiled with the 8-wave geometry");
  // (one table for every kernel: the 8-wave kernels hop block by block, 512 outputs; the 12-wave kernels
  // hop once per pair of their blocks of 256: sim_types.h RNG_HOP)
  const std::vector<uint64_t> jump = modle_host::build_jump_table(512);
  std::vector<f64> zig;
  zig.insert(zig.end(), ZIG_NORM_X, ZIG_NORM_X + 129);
  zig.insert(zig.end(), ZIG_NORM_Y, ZIG_NORM_Y + 129);
  zig.insert(zig.end(), ZIG_EXP_X, ZIG_EXP_X + 257);
  zig.insert(zig.end(), ZIG_EXP_Y, ZIG_EXP_Y + 257);
  if (h->d_jump.ensure(jump.size()) != hipSuccess ||
      h->d_zig.ensure(zig.size()) != hipSuccess ||
      h->d_counter.ensure(1) != hipSuccess || h->d_phase_out.ensure(2) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&h->h_abort), 64, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_abort), h->h_abort, 0) != hipSuccess ||
      hipMemcpy(h->d_jump.p, jump.data(), jump.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->d_zig.p, zig.data(), zig.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipEventCreate(&h->ev_start) != hipSuccess || hipEventCreate(&h->ev_stop) != hipSuccess) {
    set_err(err, errlen, "device allocation failed");
    return nullptr;
  }
  return h.release();
}

void modle_hip_destroy(modle_hip_handle* h) {
  if (h == nullptr) return;
  (void)hipSetDevice(h->device);
  if (h->in_flight) {
    // A launch is still in flight (the caller is going down on an error path, or the interpreter is
    // shutting down): raise the abort word and give the kernel the drain time -- never an unbounded
    // wait.  A kernel that does not drain is a hung device: everything the handle owns is LEAKED
    // (freeing memory a running kernel uses waits for that kernel) and the caller gets its exit.
    {
      std::lock_guard<std::mutex> lock(h->abort_mu);
      __atomic_store_n(h->h_abort, 1u, __ATOMIC_RELEASE);
    }
    const auto give_up = std::chrono::steady_clock::now() +
                         std::chrono::duration_cast<std::chrono::steady_clock::duration>(
                             std::chrono::duration<double>(h->drain_timeout_s));
    bool drained = false;
    for (;;) {
      const hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess) {
        drained = true;
        break;
      }
      if (q != hipErrorNotReady || std::chrono::steady_clock::now() > give_up) break;
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    if (!drained) {
      (void)hipGetLastError();
      std::fprintf(stderr, "modle_hip_destroy: the launch in flight did not drain within %.0f s of the abort word; "
                           "the handle's device memory is leaked (hung device: exit the process)\n",
                   h->drain_timeout_s);
      return;  // (no frees, no event destroys, no delete: each of them would wait for the kernel)
    }
    h->in_flight = false;
  }
  if (h->ev_start != nullptr) (void)hipEventDestroy(h->ev_start);
  if (h->ev_stop != nullptr) (void)hipEventDestroy(h->ev_stop);
  if (h->h_abort != nullptr) (void)hipHostFree(h->h_abort);
  if (h->h_remaining != nullptr) (void)hipHostFree(h->h_remaining);
  if (h->trace_host != nullptr) {
    (void)hipHostUnregister(h->trace_host);
    ::munmap(h->trace_host, h->trace_bytes);
  }
  delete h;
}

int modle_hip_reset(modle_hip_handle* h) {
  if (h == nullptr) return MODLE_HIP_ERR_ARG;
  if (h->in_flight) return MODLE_HIP_ERR_STATE;
  (void)hipSetDevice(h->device);
  h->intervals.clear();
  h->launch_map.clear();
  return MODLE_HIP_OK;
}

int modle_hip_add_interval(modle_hip_handle* h, uint64_t start, uint64_t end,
                           const uint64_t* bar_pos, const uint8_t* bar_dir,
                           const double* bar_stp_active, const double* bar_stp_inactive,
                           size_t n_barriers, void* d_contacts, void* d_occupancy, char* err,
                           size_t errlen) {
  if (h == nullptr) return MODLE_HIP_ERR_ARG;
  if (h->in_flight) {
    set_err(err, errlen, "a launch is in flight");
    return MODLE_HIP_ERR_STATE;
  }
  if (const char* msg = modle_host::check_limits(h->cfg, start, end, 1, n_barriers)) {
    set_err(err, errlen, msg);
    return MODLE_HIP_ERR_ARG;
  }
  if (n_barriers != 0 && (bar_pos == nullptr || bar_dir == nullptr || bar_stp_active == nullptr ||
                          bar_stp_inactive == nullptr)) {
    set_err(err, errlen, "null barrier arrays");
    return MODLE_HIP_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(h->device));
  auto rec = std::make_unique<IntervalRec>();
  rec->start = start;
  rec->end = end;
  rec->n_barriers = n_barriers;
  modle_hip_matrix_shape(&h->cfg, end - start, &rec->nrows, &rec->ncols);
  // The reference sorts the barriers of every task by position (State::operator=,
  // simulation.cpp:741-761 -> ExtrusionBarriers::sort, extrusion_barriers.cpp:237-257); here
  // that happens once per interval, so callers may pass them in BED order.  Barrier indices in
  // collision words refer to the sorted order.
  std::vector<uint64_t> spos(bar_pos, bar_pos + n_barriers);
  std::vector<u8> dir(bar_dir, bar_dir + n_barriers);
  std::vector<f64> sa(bar_stp_active, bar_stp_active + n_barriers);
  std::vector<f64> si(bar_stp_inactive, bar_stp_inactive + n_barriers);
  modle_hip_sort_barriers(spos.data(), dir.data(), sa.data(), si.data(), n_barriers);
  std::vector<u32> pos(n_barriers);
  std::vector<f64> stp(3 * n_barriers);
  for (size_t i = 0; i < n_barriers; ++i) {
    if (spos[i] < start || spos[i] >= end) {
      set_err(err, errlen, "barriers must lie inside the interval");
      return MODLE_HIP_ERR_ARG;
    }
    if (dir[i] != MODLE_HIP_DIR_FWD && dir[i] != MODLE_HIP_DIR_REV) {
      set_err(err, errlen, "barrier direction must be MODLE_HIP_DIR_FWD or MODLE_HIP_DIR_REV");
      return MODLE_HIP_ERR_ARG;
    }
    pos[i] = static_cast<u32>(spos[i]);
    stp[i] = sa[i];
    stp[n_barriers + i] = si[i];
    stp[2 * n_barriers + i] = modle_hip_occupancy_from_stp(sa[i], si[i]);
  }
  HIP_TRY(rec->bar_pos.ensure(n_barriers));
  HIP_TRY(rec->bar_dir.ensure(n_barriers));
  HIP_TRY(rec->bar_stp.ensure(3 * n_barriers));
  HIP_TRY(rec->missed.ensure(1));
  HIP_TRY(hipMemset(rec->missed.p, 0, 8));
  const std::vector<u32> buckets = modle_host::build_barrier_buckets(start, end, pos);
  rec->n_buckets = buckets.size();
  HIP_TRY(rec->bar_bucket.ensure(buckets.size()));
  HIP_TRY(hipMemcpy(rec->bar_bucket.p, buckets.data(), buckets.size() * 4, hipMemcpyHostToDevice));
  if (n_barriers != 0) {
    HIP_TRY(hipMemcpy(rec->bar_pos.p, pos.data(), n_barriers * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(rec->bar_dir.p, dir.data(), n_barriers, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(rec->bar_stp.p, stp.data(), 3 * n_barriers * 8, hipMemcpyHostToDevice));
  }
  const size_t nwords = rec->nrows * rec->ncols + 1;
  if (d_contacts != nullptr) {
    rec->d_contacts = static_cast<u32*>(d_contacts);
  } else {
    HIP_TRY(rec->own_contacts.ensure(nwords));
    HIP_TRY(hipMemset(rec->own_contacts.p, 0, nwords * 4));
    rec->d_contacts = rec->own_contacts.p;
  }
  if (d_occupancy != nullptr) {
    rec->d_occupancy = static_cast<u64*>(d_occupancy);
  } else if (h->cfg.track_1d_lef_position) {
    HIP_TRY(rec->own_occupancy.ensure(rec->ncols));
    HIP_TRY(hipMemset(rec->own_occupancy.p, 0, rec->ncols * 8));
    rec->d_occupancy = rec->own_occupancy.p;
  }
  h->intervals.push_back(std::move(rec));
  return static_cast<int>(h->intervals.size()) - 1;
}

int modle_hip_submit_tasks(modle_hip_handle* h, int interval_id, const modle_hip_task* tasks,
                           size_t n_tasks, char* err, size_t errlen) {
  if (h == nullptr || interval_id < 0 || static_cast<size_t>(interval_id) >= h->intervals.size() ||
      (tasks == nullptr && n_tasks != 0)) {
    set_err(err, errlen, "invalid arguments");
    return MODLE_HIP_ERR_ARG;
  }
  if (h->in_flight) {
    set_err(err, errlen, "a launch is in flight");
    return MODLE_HIP_ERR_STATE;
  }
  IntervalRec& rec = *h->intervals[static_cast<size_t>(interval_id)];
  for (size_t i = 0; i < n_tasks; ++i) {
    if (const char* msg = modle_host::check_limits(h->cfg, rec.start, rec.end, tasks[i].num_lefs,
                                                   rec.n_barriers)) {
      set_err(err, errlen, msg);
      return MODLE_HIP_ERR_ARG;
    }
    rec.pending.push_back(tasks[i]);
  }
  return MODLE_HIP_OK;
}

// The workspace's placement (round 5; profiles/r05zr/workspace_placement_probe.txt).  The same 1 - 3 GB allocation runs
// the 12-wave kernels at one of about three speeds, up to 6 % apart (the 8-wave kernels: 2 %), depending on which
// physical pages the driver handed out: every new allocation is a new draw -- the same virtual address behind a hole 4 KiB
// larger lands on another level -- and nothing at the HIP level chooses pages.  But the levels show: a streaming probe
// with the launch's geometry (every wave reads and writes through its own slot, no arithmetic; sim_kernels.hip) takes
// 0.605 / 0.66 / 0.69 ms on the three of them and told which speed the next launch would run at in 100 % of the
// cases measured.  So a workspace that has to be allocated is allocated up to `tries` times, each candidate behind a
// small hole of another size, probed (three repetitions, the fastest counts), and the best one is kept; the search
// makes half of its draws in any case and ends early after that once a placement 11 % better than the worst has been
// seen.  A draw costs 2.3 ms and is fast one
// time in three on some boxes and one time in six on others (profiles/r05zr/ws_search_strategies_last_run.txt: holding
// the losers or other hole sizes change nothing), hence 24: at most 55 ms, once per handle and workspace size.
// MODLE_HIP_WORKSPACE_TRIES=1 switches the search off.
static hipError_t place_workspace(modle_hip_handle* h, size_t slot_stride, int grid, int waves) {
  const size_t bytes = slot_stride * static_cast<size_t>(grid) * static_cast<size_t>(waves);
  if (bytes <= h->d_workspace.n) return hipSuccess;  // (large enough, and chosen when it was allocated)
  h->d_workspace.reset();
  h->ws_tries = 0;
  h->ws_probe_ms = h->ws_probe_worst_ms = 0.0f;
  hipError_t e = h->d_workspace.ensure(bytes);
  if (e != hipSuccess) return e;
  int tries = 24;
  if (const char* t = std::getenv("MODLE_HIP_WORKSPACE_TRIES"); t != nullptr && std::atoi(t) >= 1) tries = std::min(std::atoi(t), 64);
  if (tries <= 1) return hipSuccess;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    if (e0 != nullptr) (void)hipEventDestroy(e0);
    (void)hipGetLastError();
    return hipSuccess;  // (no probe: the first allocation it is)
  }
  const size_t array_stride = std::max<size_t>((slot_stride / 28) & ~size_t(255), 256);
  const u32 hot = static_cast<u32>(std::min<size_t>(9216, array_stride)) & ~15u;
  const u32 n_arrays = static_cast<u32>(std::min<size_t>(24, slot_stride / array_stride));
  const auto probe = [&](char* p) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      float ms = 1e30f;
      if (hipEventRecord(e0, h->stream) != hipSuccess) break;
      modle_launch::probe_workspace(p, slot_stride, static_cast<u32>(grid), static_cast<u32>(waves), array_stride, n_arrays, hot, 4,
                                    h->stream);
      if (hipEventRecord(e1, h->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
          hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
        break;
      best = std::min(best, ms);
    }
    return best;
  };
  float best = probe(h->d_workspace.p), worst = best;
  h->ws_tries = 1;
  std::vector<DevBuf<char>> holes(static_cast<size_t>(tries));  // (kept until the search is over: a freed hole would be handed out again)
  for (int t = 1; t < tries && best < 1e29f; ++t) {
    DevBuf<char> cand;
    if (holes[static_cast<size_t>(t)].ensure((static_cast<size_t>(t) * 3 + 1) * (260u << 10)) != hipSuccess ||
        cand.ensure(bytes) != hipSuccess) {
      (void)hipGetLastError();  // (out of memory for a second candidate: the best so far it is)
      break;
    }
    const float ms = probe(cand.p);
    ++h->ws_tries;
    worst = std::max(worst, ms);
    if (ms < best) {
      best = ms;
      h->d_workspace.swap(cand);
    }
    // (the levels are not three clean steps: 0.604 / 0.615 / 0.627 / 0.655 / 0.69 ms have all been kept, and the launch
    // that followed took 5 620 / 5 680 / 5 757 / 5 820 / 5 950 ms -- so at least half of the draws are made whatever
    // has been seen, and the rest only while no candidate is 11 % better than the worst)
    if (t + 1 >= tries / 2 && best <= 0.89f * worst) break;
  }
  h->ws_probe_ms = best < 1e29f ? best : 0.0f;
  h->ws_probe_worst_ms = worst < 1e29f ? worst : 0.0f;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipGetLastError();
  return hipSuccess;
}

int modle_hip_launch(modle_hip_handle* h, void* stream, char* err, size_t errlen) {
  if (h == nullptr) return MODLE_HIP_ERR_ARG;
  if (h->in_flight) {
    set_err(err, errlen, "a launch is already in flight");
    return MODLE_HIP_ERR_STATE;
  }
  HIP_TRY(hipSetDevice(h->device));
  h->stream = static_cast<hipStream_t>(stream);
  // Gather pending tasks, largest chromosomes first (long tasks must not start last).  Nothing
  // of the handle's bookkeeping changes until the kernel has been enqueued: a launch that fails
  // on the way leaves every task pending and every result slot as it was.
  std::vector<Task> tasks;
  std::vector<std::pair<int, size_t>> launch_map;
  u32 max_lefs = 1, max_barriers = 0;
  for (size_t iv = 0; iv < h->intervals.size(); ++iv) {
    const IntervalRec& rec = *h->intervals[iv];
    const size_t base = rec.results.size();
    for (size_t k = 0; k < rec.pending.size(); ++k) {
      const modle_hip_task& t = rec.pending[k];
      Task d;
      d.interval = static_cast<u32>(iv);
      d.num_lefs = static_cast<u32>(t.num_lefs);
      d.cell_id = t.cell_id;
      d.num_target_epochs = t.num_target_epochs;
      d.num_target_contacts = t.num_target_contacts;
      d.contacts_per_epoch = modle_hip_compute_contacts_per_epoch(&h->cfg, t.num_lefs);
      std::memcpy(d.prng, t.prng, sizeof(d.prng));
      tasks.push_back(d);
      launch_map.emplace_back(static_cast<int>(iv), base + k);
      max_lefs = std::max(max_lefs, d.num_lefs);
    }
    if (!rec.pending.empty()) max_barriers = std::max<u32>(max_barriers, static_cast<u32>(rec.n_barriers));
  }
  if (tasks.empty()) {
    h->n_launched = 0;
    h->launch_map.clear();
    return MODLE_HIP_OK;
  }
  std::vector<size_t> order(tasks.size());
  std::iota(order.begin(), order.end(), size_t(0));
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
    const IntervalRec& ia = *h->intervals[tasks[a].interval];
    const IntervalRec& ib = *h->intervals[tasks[b].interval];
    const uint64_t wa = static_cast<uint64_t>(tasks[a].num_lefs) + ia.n_barriers;
    const uint64_t wb = static_cast<uint64_t>(tasks[b].num_lefs) + ib.n_barriers;
    return wa > wb;
  });
  std::vector<Task> sorted(tasks.size());
  std::vector<std::pair<int, size_t>> sorted_map(tasks.size());
  for (size_t i = 0; i < order.size(); ++i) {
    sorted[i] = tasks[order[i]];
    sorted_map[i] = launch_map[order[i]];
  }

  std::vector<Interval> ivs(h->intervals.size());
  for (size_t iv = 0; iv < h->intervals.size(); ++iv) {
    const IntervalRec& rec = *h->intervals[iv];
    Interval& d = ivs[iv];
    d.start = static_cast<u32>(rec.start);
    d.end = static_cast<u32>(rec.end);
    d.n_barriers = static_cast<u32>(rec.n_barriers);
    d.pad_ = 0;
    d.bar_pos = rec.bar_pos.p;
    d.bar_dir = rec.bar_dir.p;
    d.bar_stp_active = rec.bar_stp.p;
    d.bar_stp_inactive = rec.bar_stp.p + rec.n_barriers;
    d.bar_occupancy = rec.bar_stp.p + 2 * rec.n_barriers;
    d.contacts = rec.d_contacts;
    d.occupancy_1d = rec.d_occupancy;
    d.missed_updates = rec.missed.p;
    d.nrows = rec.nrows;
    d.ncols = rec.ncols;
    d.bar_bucket = rec.bar_bucket.p;
    d.bucket_shift = BAR_BUCKET_SHIFT;
    d.n_buckets = static_cast<u32>(rec.n_buckets);
  }
  // one workgroup per CU even when there are fewer tasks than waves: the waves that win a task
  // are then spread over all CUs instead of being packed 8 to a CU
  int grid = std::max(1, static_cast<int>(std::min<size_t>(h->num_cus, sorted.size())));
  if (const char* g = std::getenv("MODLE_HIP_GRID"); g != nullptr && std::atoi(g) >= 1) {
    // diagnostic: fewer workgroups than CUs (what do the waves of a CU share, what do the CUs share?)
    grid = std::min(grid, std::atoi(g));
  }
  const auto layout = modle_host::workspace_layout(max_lefs, max_barriers, h->params.hist_len);
  // Waves per workgroup (round 5): 12 -- three per SIMD, 168 VGPRs, 256-output PRNG blocks, 256-key LDS buffers -- is
  // worth 1.2 % on a launch that fills the slots and whose epochs re-insert few units (the default parameters: 70 - 134
  // per epoch on chr1); a launch whose epochs re-insert more than the small buffers hold (BASELINE configs[4]: 600 - 800)
  // loses 17 % with them, and the fixed roles of a launch that leaves slots empty are laid out for 8.  Expected
  // re-insertions per epoch and direction ~ LEFs x release probability.  MODLE_HIP_WAVES=8 / 12 forces it.
  int waves = 8;
  {
    const bool paired_launch = sorted.size() <= static_cast<size_t>(grid) * 4;
    const double p_rel = std::max(h->params.p_release, h->params.p_release_burnin);
    if (!paired_launch && sorted.size() >= static_cast<size_t>(grid) * 12 && static_cast<double>(max_lefs) * p_rel < 170.0)
      waves = 12;
    if (const char* e = std::getenv("MODLE_HIP_WAVES"); e != nullptr && (std::atoi(e) == 8 || std::atoi(e) == 12))
      waves = std::atoi(e);
    // (helper-wave mode asked for by name -- tests, A/B runs -- means the kernels that have it)
    if (const char* pm = std::getenv("MODLE_HIP_PAIRED"); pm != nullptr && pm[0] != '\0' && pm[0] != '0') waves = 8;
#if defined(MODLE_EXP_LDS_WS) || defined(MODLE_STAGE_TRACE) || (MODLE_WAVES_PER_CU != 8)
    waves = 8;  // (measurement / diagnostic builds)
#endif
  }
  const size_t n_slots = static_cast<size_t>(grid) * static_cast<size_t>(waves);
#if defined(MODLE_EXP_LDS_WS) && !defined(MODLE_EXP_LDS_WS_OFF)
  if (layout.u32_words * 4 + layout.u8_bytes + layout.hit_words * 4 > static_cast<size_t>(MODLE_EXP_LDS_WS)) {
    set_err(err, errlen, "measurement build: the cell's state does not fit the LDS slice (" +
                             std::to_string(layout.u32_words * 4 + layout.u8_bytes + layout.hit_words * 4) + " bytes)");
    return MODLE_HIP_ERR_ARG;
  }
#endif
  HIP_TRY(h->d_intervals.ensure(ivs.size()));
  HIP_TRY(h->d_tasks.ensure(sorted.size()));
  HIP_TRY(h->d_results.ensure(sorted.size()));
  HIP_TRY(h->d_status.ensure(sorted.size()));
#ifdef MODLE_EXP_REALLOC  // (measurement: does the placement of the workspace decide which of a box's two speeds a process runs at?)
  {
    static DevBuf<char> hole;  // a hole of a different size in front of every new workspace
    static int launches = 0;
    h->d_workspace.reset();
    hole.reset();
    (void)hole.ensure(static_cast<size_t>(1 + (launches++ * 37) % 200) << 20);
  }
#endif
  HIP_TRY(place_workspace(h, layout.total_bytes, grid, waves));
  if (const char* e = std::getenv("MODLE_HIP_POISON_WORKSPACE"); e != nullptr && e[0] != '\0' && e[0] != '0') {
    // tests: a cell must not depend on what its workspace slot held before (fresh device memory is usually zero, the
    // placement probe and earlier cells leave anything): every byte 0xA5 before the launch
    HIP_TRY(hipMemsetAsync(h->d_workspace.p, 0xA5, layout.total_bytes * n_slots, h->stream));
  }
  if (h->remaining_cap < ivs.size()) {
    if (h->h_remaining != nullptr) (void)hipHostFree(h->h_remaining);
    h->h_remaining = nullptr;
    h->remaining_cap = 0;
    void* hp = nullptr;
    HIP_TRY(hipHostMalloc(&hp, std::max<size_t>(ivs.size(), 64) * 4, hipHostMallocMapped));
    void* dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
    h->h_remaining = static_cast<u32*>(hp);
    h->d_remaining = static_cast<u32*>(dp);
    h->remaining_cap = std::max<size_t>(ivs.size(), 64);
  }
  for (size_t iv = 0; iv < ivs.size(); ++iv) h->h_remaining[iv] = 0;
  for (const Task& t : sorted) h->h_remaining[t.interval] += 1;
  HIP_TRY(hipMemcpyAsync(h->d_intervals.p, ivs.data(), ivs.size() * sizeof(Interval),
                         hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_tasks.p, sorted.data(), sorted.size() * sizeof(Task),
                         hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemsetAsync(h->d_counter.p, 0, 4, h->stream));
  HIP_TRY(hipMemsetAsync(h->d_status.p, 0xFF, sorted.size() * 4, h->stream));
  // the source vectors must outlive the async copies
  HIP_TRY(hipStreamSynchronize(h->stream));

  SimArgs a;
  a.params = h->params;
#ifdef MODLE_EXP_SWITCH
  a.params.exp_flags = std::getenv("MODLE_HIP_EXP") != nullptr ? static_cast<u32>(std::atoi(std::getenv("MODLE_HIP_EXP"))) : 0u;
  a.params.exp_pad_ = 0;
#endif
  a.tables.jump = h->d_jump.p;
  a.tables.zig = h->d_zig.p;
  a.intervals = h->d_intervals.p;
  a.tasks = h->d_tasks.p;
  a.results = h->d_results.p;
  a.status = h->d_status.p;
  a.task_counter = h->d_counter.p;
  a.abort_flag = h->d_abort;
  a.interval_remaining = h->d_remaining;
  a.trace = nullptr;
  a.trace_cap = 0;
  a.pad2_ = 0;
  if (const char* shm = std::getenv("MODLE_HIP_TRACE_SHM"); shm != nullptr) {
    // diagnostic: the trace of task 0 goes to a file-backed, host-coherent mapping so that it
    // survives a GPU fault that aborts the process (mapped and registered once per handle)
    constexpr u32 kTraceEpochs = 4096;
    const size_t bytes = static_cast<size_t>(kTraceEpochs) * TRACE_STAGES * TRACE_WORDS_PER_STAGE * 8;
    if (h->trace_host == nullptr) {
      const int fd = ::open(shm, O_RDWR | O_CREAT | O_TRUNC, 0644);
      if (fd >= 0 && ::ftruncate(fd, static_cast<off_t>(bytes)) == 0) {
        void* hp = ::mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        void* dp = nullptr;
        if (hp != MAP_FAILED) {
          if (hipHostRegister(hp, bytes, hipHostRegisterMapped) == hipSuccess &&
              hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            h->trace_host = hp;
            h->trace_bytes = bytes;
            h->trace_dev = static_cast<u64*>(dp);
          } else {
            ::munmap(hp, bytes);
          }
        }
      }
      if (fd >= 0) ::close(fd);
    }
    if (h->trace_dev != nullptr) {
      a.trace = h->trace_dev;
      a.trace_cap = kTraceEpochs;
    }
  }
  a.state_log = nullptr;
  a.state_log_cap = 0;
  a.pad3_ = 0;
  if (h->state_log_cap != 0) {
    const size_t words = sorted.size() * static_cast<size_t>(h->state_log_cap) * STATE_LOG_WORDS;
    HIP_TRY(h->d_state_log.ensure(words));
    HIP_TRY(hipMemsetAsync(h->d_state_log.p, 0xFF, words * 8, h->stream));
    a.state_log = h->d_state_log.p;
    a.state_log_cap = h->state_log_cap;
  }
  a.phase_ticks = nullptr;
#ifdef MODLE_PHASE_TIMERS
  HIP_TRY(h->d_phase_ticks.ensure(20 + 3 * sorted.size()));
  HIP_TRY(hipMemsetAsync(h->d_phase_ticks.p, 0, (20 + 3 * sorted.size()) * 8, h->stream));
  HIP_TRY(hipMemsetAsync(h->d_phase_ticks.p + 18, 0xFF, 8, h->stream));
  a.phase_ticks = h->d_phase_ticks.p;
#endif
  a.workspace = h->d_workspace.p;
  a.workspace_stride = layout.total_bytes;
  a.n_tasks = static_cast<u32>(sorted.size());
  a.max_lefs = max_lefs;
  a.max_barriers = max_barriers;
  // helper-wave mode for launches that leave at least half of the wave slots empty (sim_pair.h);
  // MODLE_HIP_PAIRED=0 / 1 turns it off / on whatever the number of tasks (tests, A/B runs)
  a.pair_mains = 0;
  {
    constexpr size_t kMaxMains = 4;  // (of the 8 waves of the kernels that know fixed roles)
    bool paired = waves == 8 && sorted.size() <= static_cast<size_t>(grid) * kMaxMains;
    if (const char* pm = std::getenv("MODLE_HIP_PAIRED"); pm != nullptr && pm[0] != '\0') paired = pm[0] != '0';
#ifdef MODLE_STAGE_TRACE
    paired = false;  // (the stage trace records the generator position between the phases)
#endif
    if (paired && waves == 8)
      a.pair_mains = static_cast<u32>(std::min(kMaxMains, (sorted.size() + static_cast<size_t>(grid) - 1) / grid));
  }
  a.tail_helpers = a.pair_mains == 0 ? 1u : 0u;
  if (const char* th = std::getenv("MODLE_HIP_TAIL_HELPERS"); th != nullptr && th[0] == '0') a.tail_helpers = 0;
#ifdef MODLE_STAGE_TRACE
  a.tail_helpers = 0;
#endif
  a.test_fault = 0;
  if (const char* tf = std::getenv("MODLE_HIP_TEST_FAULT"); tf != nullptr && std::strcmp(tf, "stuck_helper") == 0)
    a.test_fault = TEST_FAULT_STUCK_HELPER;  // (tests/test_gpu_wait_deadline.py)
  a.active_waves = static_cast<u32>(waves);
  if (const char* aw = std::getenv("MODLE_HIP_ACTIVE_WAVES"); aw != nullptr) {
    // diagnostic: how the kernel time scales with the waves in flight per CU
    const int v = std::atoi(aw);
    if (v >= 1 && v <= waves) a.active_waves = static_cast<u32>(v);
  }
  {
    // The abort word is cleared, the kernel enqueued and the launch marked as in flight under the
    // lock modle_hip_cancel takes: a cancel from another thread either finds nothing in flight
    // (it came before this launch) or raises the word after the reset -- never in between, where
    // the reset would swallow it.
    std::lock_guard<std::mutex> lock(h->abort_mu);
    __atomic_store_n(h->h_abort, 0u, __ATOMIC_RELEASE);
    h->cancelled = false;
    const bool ev0 = hipEventRecord(h->ev_start, h->stream) == hipSuccess;
    const bool wide = modle_hip_size_class(&h->cfg, max_lefs) != 0;
    if (waves == 12) {
      if (wide) simulate_wide12(grid, h->stream, a); else simulate_narrow12(grid, h->stream, a);
    } else {
      if (wide) simulate_wide(grid, h->stream, a); else simulate_narrow(grid, h->stream, a);
    }
    HIP_TRY(hipGetLastError());
    // The kernel is enqueued: commit the bookkeeping NOW.  Nothing after this point may make the
    // call fail -- a caller that sees an error retries, and the same tasks would then be simulated
    // twice into the same matrices.  A failed event record only costs the timing of this launch.
    for (auto& recp : h->intervals) {
      IntervalRec& rec = *recp;
      rec.results.resize(rec.results.size() + rec.pending.size());
      rec.pending.clear();
    }
    h->launch_map.swap(sorted_map);
    h->n_launched = sorted.size();
    h->in_flight = true;
    h->launched_at = std::chrono::steady_clock::now();
    h->last_launch.n_tasks = sorted.size();
    h->last_launch.num_cus = static_cast<uint64_t>(h->num_cus);
    h->last_launch.workgroups = static_cast<uint64_t>(grid);
    h->last_launch.waves_per_workgroup = static_cast<uint64_t>(waves);
    h->last_launch.main_waves_per_workgroup = a.pair_mains != 0 ? a.pair_mains : a.active_waves;
    h->last_launch.helper_waves = a.pair_mains != 0 ? 1 : 0;
#ifdef MODLE_RNG_PHILOX
    h->last_launch.prng_producer_waves = 0;
#else
    h->last_launch.prng_producer_waves = (a.pair_mains != 0 && a.pair_mains <= 2) ? 1 : 0;
#endif
    h->last_launch.tail_helpers = a.tail_helpers;
    h->last_launch.size_class = wide ? 1 : 0;
    h->last_launch.workspace_tries = h->ws_tries;
    h->last_launch.workspace_probe_us = static_cast<uint64_t>(h->ws_probe_ms * 1000.0f + 0.5f);
    h->last_launch.workspace_probe_worst_us = static_cast<uint64_t>(h->ws_probe_worst_ms * 1000.0f + 0.5f);
    const bool ev1 = hipEventRecord(h->ev_stop, h->stream) == hipSuccess;
    h->timing_valid = ev0 && ev1;
    if (!h->timing_valid) (void)hipGetLastError();  // (clears the sticky error of the failed record)
  }
  return MODLE_HIP_OK;
}

int modle_hip_cancel(modle_hip_handle* h, char* err, size_t errlen) {
  if (h == nullptr) return MODLE_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lock(h->abort_mu);
  if (!h->in_flight) return MODLE_HIP_OK;
  HIP_TRY(hipSetDevice(h->device));
  // every wave reads the word at the top of its next epoch and stops pulling tasks
  // (a plain store into host-mapped memory: see the handle's h_abort; found by
  // tests/test_gpu_cancel.py, which saw a cancel issued through a stream take effect only when
  // the first workgroups of the launch retired)
  __atomic_store_n(h->h_abort, 1u, __ATOMIC_RELEASE);
  h->cancelled = true;
  return MODLE_HIP_OK;
}

int modle_hip_set_wait_timeout(modle_hip_handle* h, double seconds) {
  if (h == nullptr || !(seconds > 0.0)) return MODLE_HIP_ERR_ARG;
  h->wait_timeout_s = seconds;
  return MODLE_HIP_OK;
}

int modle_hip_runtime_versions(int* built_with, int* runtime) {
  if (built_with == nullptr || runtime == nullptr) return MODLE_HIP_ERR_ARG;
  *built_with = HIP_VERSION;
  *runtime = 0;
  return hipRuntimeGetVersion(runtime) == hipSuccess ? MODLE_HIP_OK : MODLE_HIP_ERR_DEVICE;
}

int modle_hip_last_launch_info(modle_hip_handle* h, modle_hip_launch_info* info) {
  if (h == nullptr || info == nullptr) return MODLE_HIP_ERR_ARG;
  *info = h->last_launch;
  return MODLE_HIP_OK;
}

int modle_hip_wait(modle_hip_handle* h, char* err, size_t errlen) {
  if (h == nullptr) return MODLE_HIP_ERR_ARG;
  if (!h->in_flight) return MODLE_HIP_OK;
  HIP_TRY(hipSetDevice(h->device));
  // Bounded wait.  The stream is polled (no HIP call waits with a deadline); when the launch has
  // been running for longer than the deadline the abort word is raised -- every wave reads it at the
  // top of every sixteenth epoch and in every spin loop of the hand-over protocols (sim_rng.h:
  // spin_nap_aborted) -- and the kernel gets `drain_timeout_s` to drain.  A launch that drains is
  // reported as MODLE_HIP_ERR_TIMEOUT with the handle usable again (modle_hip_reset); one that does
  // not is a hung device: MODLE_HIP_ERR_DEVICE, the launch stays in flight and the process should
  // exit (never re-exec a process that has touched the GPU).
  bool timed_out = false;
  {
    using clock = std::chrono::steady_clock;
    const bool has_deadline = h->wait_timeout_s > 0.0;
    const auto deadline = h->launched_at + std::chrono::duration_cast<clock::duration>(
                                               std::chrono::duration<double>(has_deadline ? h->wait_timeout_s : 0.0));
    clock::time_point drain_deadline{};
    unsigned polls = 0;
    for (;;) {
      const hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) {
        set_err(err, errlen, std::string("hipStreamQuery: ") + hipGetErrorString(q));
        return MODLE_HIP_ERR_DEVICE;
      }
      const auto now = clock::now();
      if (!timed_out && has_deadline && now > deadline) {
        std::lock_guard<std::mutex> lock(h->abort_mu);
        __atomic_store_n(h->h_abort, 1u, __ATOMIC_RELEASE);
        timed_out = true;
        drain_deadline = now + std::chrono::duration_cast<clock::duration>(
                                   std::chrono::duration<double>(h->drain_timeout_s));
      } else if (timed_out && now > drain_deadline) {
        set_err(err, errlen, "the launch exceeded the wait deadline of " + std::to_string(h->wait_timeout_s) +
                                 " s and did not drain within " + std::to_string(h->drain_timeout_s) +
                                 " s of the abort word being raised: the device is hung");
        return MODLE_HIP_ERR_DEVICE;
      }
      // short launches (tests) end within the first polls; long ones are polled every 200 us
      if (++polls < 4096) {
        std::this_thread::yield();
      } else {
        std::this_thread::sleep_for(std::chrono::microseconds(200));
      }
    }
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->in_flight = false;
  if (!h->timing_valid || hipEventElapsedTime(&h->last_ms, h->ev_start, h->ev_stop) != hipSuccess) {
    (void)hipGetLastError();
    h->last_ms = std::nanf("");  // (an event could not be recorded: the launch itself is fine)
  }
#ifdef MODLE_PHASE_TIMERS
  {
    static const char* names[16] = {"burnin_stats", "bind", "rank_rev", "rank_fwd", "sample", "gen_moves",
                                    "adjust_moves", "barriers+clear", "boundaries", "lef_bar", "primary",
                                    "secondary", "fix_secondary", "extrude_release", "lef_activation", "cell_total"};
    u64 ticks[20];
    HIP_TRY(hipMemcpy(ticks, h->d_phase_ticks.p, sizeof(ticks), hipMemcpyDeviceToHost));
    u64 total = 0;
    for (int i = 0; i < 14; ++i) total += ticks[i];
    std::fprintf(stderr, "[modle_hip prof] kernel %.1f ms, wave-time per phase (sum over waves, 100 MHz ticks):\n", h->last_ms);
    for (int i = 0; i < 16; ++i)
      if (i < 14 || ticks[i] != 0)  // (14, 15: free for a measurement inside a phase)
      std::fprintf(stderr, "  %-16s %12.3f s  %5.1f %%\n", names[i], static_cast<double>(ticks[i]) * 1e-8,
                   total ? 100.0 * static_cast<double>(ticks[i]) / static_cast<double>(total) : 0.0);
    std::fprintf(stderr, "  main waves with work: sum %.3f s, longest %.3f s, shortest %.3f s\n",
                 static_cast<double>(ticks[16]) * 1e-8, static_cast<double>(ticks[17]) * 1e-8,
                 static_cast<double>(ticks[18]) * 1e-8);
  }
#endif
  std::vector<CellResult> res(h->n_launched);
  std::vector<u32> status(h->n_launched);
  HIP_TRY(hipMemcpy(res.data(), h->d_results.p, res.size() * sizeof(CellResult),
                    hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(status.data(), h->d_status.p, status.size() * 4, hipMemcpyDeviceToHost));
#ifdef MODLE_PHASE_TIMERS
  if (const char* path = std::getenv("MODLE_PROF_TASK_TIMES")) {
    std::vector<u64> tt(3 * h->n_launched);
    HIP_TRY(hipMemcpy(tt.data(), h->d_phase_ticks.p + 20, tt.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = std::fopen(path, "w")) {
      // (queue order: interval, start, end in ticks, wave slot, epochs, burn-in epochs)
      for (size_t i = 0; i < h->n_launched; ++i)
        std::fprintf(f, "%d %llu %llu %llu %llu %llu\n", h->launch_map[i].first, (unsigned long long)tt[3 * i],
                     (unsigned long long)tt[3 * i + 1], (unsigned long long)tt[3 * i + 2],
                     (unsigned long long)res[i].epochs, (unsigned long long)res[i].burnin_epochs);
      std::fclose(f);
    }
  }
#endif
  int rc = MODLE_HIP_OK;
  bool any_cancelled = false;
  for (size_t i = 0; i < res.size(); ++i) {
    const auto [iv, idx] = h->launch_map[i];
    static_assert(sizeof(CellResult) == sizeof(modle_hip_cell_result), "result layouts differ");
    std::memcpy(&h->intervals[static_cast<size_t>(iv)]->results[idx], &res[i], sizeof(CellResult));
    if (status[i] == ERR_CANCELLED) {
      any_cancelled = true;
      if (rc == MODLE_HIP_OK) {
        set_err(err, errlen, "the launch was cancelled (modle_hip_cancel)");
        rc = MODLE_HIP_ERR_CANCELLED;
      }
    } else if (status[i] != 0 && (rc == MODLE_HIP_OK || rc == MODLE_HIP_ERR_CANCELLED)) {
      set_err(err, errlen,
              "task " + std::to_string(i) + " failed on the device with status " +
                  std::to_string(status[i]) + " (internal capacity exceeded)");
      rc = MODLE_HIP_ERR_STATE;
    }
  }
  // (a launch that finished on its own just as the deadline passed has no cancelled task: its outputs are
  // complete and it is reported as what it is)
  if (timed_out && any_cancelled) {
    set_err(err, errlen, "the launch exceeded the wait deadline of " + std::to_string(h->wait_timeout_s) +
                             " s (MODLE_HIP_WAIT_TIMEOUT_S / modle_hip_set_wait_timeout): it was aborted and has "
                             "drained; its outputs are incomplete");
    return MODLE_HIP_ERR_TIMEOUT;
  }
  return rc;
}

int modle_hip_enable_state_log(modle_hip_handle* h, uint32_t max_epochs_per_task, char* err,
                                size_t errlen) {
  if (h == nullptr) return MODLE_HIP_ERR_ARG;
#ifndef MODLE_STATE_LOG
  if (max_epochs_per_task != 0) {
    set_err(err, errlen,
            "this build of the library does not log the model's internal state: load "
            "libmodle_hip_statelog.so (make -C modle_amd/csrc statelog)");
    return MODLE_HIP_ERR_UNSUPPORTED;
  }
#endif
  if (h->in_flight) return MODLE_HIP_ERR_STATE;
  h->state_log_cap = max_epochs_per_task;
  return MODLE_HIP_OK;
}

int modle_hip_get_state_log(modle_hip_handle* h, int interval_id, size_t task_index,
                            uint64_t* records, size_t max_epochs, size_t* n_epochs, char* err,
                            size_t errlen) {
  if (h == nullptr || interval_id < 0 || static_cast<size_t>(interval_id) >= h->intervals.size() ||
      n_epochs == nullptr) {
    set_err(err, errlen, "invalid arguments");
    return MODLE_HIP_ERR_ARG;
  }
  if (h->in_flight || h->state_log_cap == 0 || h->d_state_log.p == nullptr) {
    set_err(err, errlen, "no state log available (enable it before the launch, wait for the launch)");
    return MODLE_HIP_ERR_STATE;
  }
  HIP_TRY(hipSetDevice(h->device));
  // the record block of the task: tasks of the LAST launch, found through the launch map
  // (launch slot -> (interval, submission index)); `task_index` counts this interval's tasks of
  // that launch in submission order
  std::vector<std::pair<size_t, size_t>> mine;  // (submission index, launch slot)
  for (size_t i = 0; i < h->launch_map.size(); ++i)
    if (h->launch_map[i].first == interval_id) mine.emplace_back(h->launch_map[i].second, i);
  std::sort(mine.begin(), mine.end());
  if (task_index >= mine.size()) {
    set_err(err, errlen, "task index out of range");
    return MODLE_HIP_ERR_ARG;
  }
  const size_t slot = mine[task_index].second;
  const size_t block = static_cast<size_t>(h->state_log_cap) * STATE_LOG_WORDS;
  std::vector<uint64_t> all(block);
  HIP_TRY(hipMemcpy(all.data(), h->d_state_log.p + slot * block, block * 8, hipMemcpyDeviceToHost));
  // a record sits at its epoch's index; epochs whose move / collision phase did not run (the
  // epoch in which the contact target is reached) have none
  size_t out = 0;
  // (records == NULL counts: max_epochs does not bound the answer then)
  for (size_t e = 0; e < h->state_log_cap && (records == nullptr || out < max_epochs); ++e) {
    if (all[e * STATE_LOG_WORDS] == ~uint64_t(0)) continue;
    if (records != nullptr)
      std::memcpy(records + out * STATE_LOG_WORDS, all.data() + e * STATE_LOG_WORDS, STATE_LOG_WORDS * 8);
    ++out;
  }
  *n_epochs = out;
  return MODLE_HIP_OK;
}

int modle_hip_interval_done(modle_hip_handle* h, int interval_id) {
  if (h == nullptr || interval_id < 0 || static_cast<size_t>(interval_id) >= h->intervals.size())
    return MODLE_HIP_ERR_ARG;
  if (!h->in_flight) return 1;
  if (h->h_remaining == nullptr || static_cast<size_t>(interval_id) >= h->remaining_cap) return 0;
  const volatile u32* p = h->h_remaining + interval_id;
  return *p == 0 ? 1 : 0;
}

int modle_hip_last_kernel_ms(modle_hip_handle* h, float* ms) {
  if (h == nullptr || ms == nullptr) return MODLE_HIP_ERR_ARG;
  *ms = h->last_ms;
  return MODLE_HIP_OK;
}

int modle_hip_get_results(modle_hip_handle* h, int interval_id, modle_hip_cell_result* results,
                          size_t n_results) {
  if (h == nullptr || interval_id < 0 || static_cast<size_t>(interval_id) >= h->intervals.size())
    return MODLE_HIP_ERR_ARG;
  const IntervalRec& rec = *h->intervals[static_cast<size_t>(interval_id)];
  if (n_results > rec.results.size()) return MODLE_HIP_ERR_ARG;
  std::memcpy(results, rec.results.data(), n_results * sizeof(modle_hip_cell_result));
  return MODLE_HIP_OK;
}

int modle_hip_interval_outputs(modle_hip_handle* h, int interval_id, void** d_contacts,
                               void** d_occupancy, uint64_t* nrows, uint64_t* ncols) {
  if (h == nullptr || interval_id < 0 || static_cast<size_t>(interval_id) >= h->intervals.size())
    return MODLE_HIP_ERR_ARG;
  const IntervalRec& rec = *h->intervals[static_cast<size_t>(interval_id)];
  if (d_contacts != nullptr) *d_contacts = rec.d_contacts;
  if (d_occupancy != nullptr) *d_occupancy = rec.d_occupancy;
  if (nrows != nullptr) *nrows = rec.nrows;
  if (ncols != nullptr) *ncols = rec.ncols;
  return MODLE_HIP_OK;
}

int modle_hip_copy_outputs(modle_hip_handle* h, int interval_id, uint32_t* contacts,
                           uint64_t* missed_updates, uint64_t* occupancy, char* err,
                           size_t errlen) {
  if (h == nullptr || interval_id < 0 || static_cast<size_t>(interval_id) >= h->intervals.size()) {
    set_err(err, errlen, "invalid interval id");
    return MODLE_HIP_ERR_ARG;
  }
  if (h->in_flight) {
    set_err(err, errlen, "a launch is in flight");
    return MODLE_HIP_ERR_STATE;
  }
  HIP_TRY(hipSetDevice(h->device));
  const IntervalRec& rec = *h->intervals[static_cast<size_t>(interval_id)];
  if (contacts != nullptr)
    HIP_TRY(hipMemcpy(contacts, rec.d_contacts, (rec.nrows * rec.ncols + 1) * 4,
                      hipMemcpyDeviceToHost));
  if (missed_updates != nullptr)
    HIP_TRY(hipMemcpy(missed_updates, rec.missed.p, 8, hipMemcpyDeviceToHost));
  if (occupancy != nullptr && rec.d_occupancy != nullptr)
    HIP_TRY(hipMemcpy(occupancy, rec.d_occupancy, rec.ncols * 8, hipMemcpyDeviceToHost));
  return MODLE_HIP_OK;
}

int modle_hip_simulate_interval(modle_hip_handle* h, uint64_t start, uint64_t end,
                                const uint64_t* bar_pos, const uint8_t* bar_dir,
                                const double* bar_stp_active, const double* bar_stp_inactive,
                                size_t n_barriers, const modle_hip_task* tasks, size_t n_tasks,
                                uint32_t* contacts, uint64_t nrows, uint64_t ncols,
                                uint64_t* missed_updates, uint64_t* occupancy,
                                modle_hip_cell_result* results, char* err, size_t errlen) {
  if (h == nullptr) return MODLE_HIP_ERR_ARG;
  if (h->in_flight) {
    set_err(err, errlen, "a launch is in flight");
    return MODLE_HIP_ERR_STATE;
  }
  for (const auto& r : h->intervals) {
    if (!r->pending.empty()) {
      // the one-call form launches every pending task of the handle: refuse to mix
      set_err(err, errlen, "tasks submitted through modle_hip_submit_tasks are still pending");
      return MODLE_HIP_ERR_STATE;
    }
  }
  const int id = modle_hip_add_interval(h, start, end, bar_pos, bar_dir, bar_stp_active,
                                        bar_stp_inactive, n_barriers, nullptr, nullptr, err,
                                        errlen);
  if (id < 0) return id;
  // the interval registered above (device matrix included: 119 MB for chr1) lives for this call
  // only; it is the last one of the handle and goes away on every return path
  struct Release {
    modle_hip_handle* h;
    ~Release() {
      (void)hipSetDevice(h->device);
      if (h->in_flight) {
        (void)hipStreamSynchronize(h->stream);
        h->in_flight = false;
      }
      h->intervals.pop_back();
      h->launch_map.clear();
      h->n_launched = 0;
    }
  } release{h};
  IntervalRec& rec = *h->intervals[static_cast<size_t>(id)];
  if (rec.nrows != nrows || rec.ncols != ncols) {
    set_err(err, errlen, "contact matrix shape does not match bin_size / diagonal_width");
    return MODLE_HIP_ERR_ARG;
  }
  int rc = modle_hip_submit_tasks(h, id, tasks, n_tasks, err, errlen);
  if (rc == MODLE_HIP_OK) rc = modle_hip_launch(h, nullptr, err, errlen);
  if (rc == MODLE_HIP_OK) rc = modle_hip_wait(h, err, errlen);
  if (rc != MODLE_HIP_OK) return rc;
  if (results != nullptr) std::memcpy(results, rec.results.data(), n_tasks * sizeof(*results));
  // accumulate into the caller's (shared) matrix like ContactMatrixDense::increment does
  const size_t nwords = nrows * ncols + 1;
  if (contacts != nullptr) {
    std::vector<uint32_t> tmp(nwords);
    HIP_TRY(hipMemcpy(tmp.data(), rec.d_contacts, nwords * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nwords; ++i) contacts[i] += tmp[i];
  }
  if (missed_updates != nullptr) {
    uint64_t m = 0;
    HIP_TRY(hipMemcpy(&m, rec.missed.p, 8, hipMemcpyDeviceToHost));
    *missed_updates += m;
  }
  if (occupancy != nullptr && rec.d_occupancy != nullptr) {
    std::vector<uint64_t> tmp(ncols);
    HIP_TRY(hipMemcpy(tmp.data(), rec.d_occupancy, ncols * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ncols; ++i) occupancy[i] += tmp[i];
  }
  return MODLE_HIP_OK;
}

int modle_hip_test_phases(modle_hip_handle* h, uint32_t phase_mask, uint64_t start, uint64_t end,
                          size_t n, uint64_t* rev_pos, uint64_t* fwd_pos, uint64_t* epoch,
                          uint64_t* rev_rank, uint64_t* fwd_rank, uint64_t* rev_moves,
                          uint64_t* fwd_moves, uint64_t* rev_coll, uint64_t* fwd_coll,
                          size_t n_barriers, const uint64_t* bar_pos, const uint8_t* bar_dir,
                          const uint8_t* bar_active, uint64_t prng[4], uint64_t* raws_consumed,
                          char* err, size_t errlen) {
  if (h == nullptr || n == 0) return MODLE_HIP_ERR_ARG;
  if (h->in_flight) return MODLE_HIP_ERR_STATE;
  HIP_TRY(hipSetDevice(h->device));
  const auto layout =
      modle_host::workspace_layout(static_cast<u32>(n), static_cast<u32>(n_barriers), 4);
  std::vector<u32> image(9 * n);
  TestImage img;
  modle_host::fill_test_image(image.data(), n, rev_pos, fwd_pos, epoch, rev_rank, fwd_rank,
                              rev_moves, fwd_moves, rev_coll, fwd_coll, img);
  // workspace image: only the barrier states need initial values
  std::vector<uint64_t> wsimg(layout.total_bytes / 8 + 1, 0);
  Workspace ws = modle_host::carve_workspace(wsimg.data(), static_cast<u32>(n),
                                             static_cast<u32>(n_barriers), 4);
  for (size_t i = 0; i < n_barriers; ++i) ws.bar_active[i] = bar_active[i];
  DevBuf<char> d_ws;
  DevBuf<u32> d_img;
  DevBuf<u32> d_bpos;
  DevBuf<u8> d_bdir;
  DevBuf<u32> d_bucket;
  HIP_TRY(d_ws.ensure(layout.total_bytes));
  HIP_TRY(d_img.ensure(9 * n));
  HIP_TRY(d_bpos.ensure(n_barriers));
  HIP_TRY(d_bdir.ensure(n_barriers));
  HIP_TRY(hipMemcpy(d_ws.p, wsimg.data(), layout.total_bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_img.p, image.data(), image.size() * 4, hipMemcpyHostToDevice));
  std::vector<u32> bp(n_barriers);
  for (size_t i = 0; i < n_barriers; ++i) bp[i] = static_cast<u32>(bar_pos[i]);
  if (n_barriers != 0) {
    HIP_TRY(hipMemcpy(d_bpos.p, bp.data(), n_barriers * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_bdir.p, bar_dir, n_barriers, hipMemcpyHostToDevice));
  }
  const std::vector<u32> buckets = modle_host::build_barrier_buckets(start, end, bp);
  HIP_TRY(d_bucket.ensure(buckets.size()));
  HIP_TRY(hipMemcpy(d_bucket.p, buckets.data(), buckets.size() * 4, hipMemcpyHostToDevice));
  PhaseArgs a;
  std::memset(&a, 0, sizeof(a));
  a.params = h->params;
  a.tables.jump = h->d_jump.p;
  a.tables.zig = h->d_zig.p;
  a.interval.start = static_cast<u32>(start);
  a.interval.end = static_cast<u32>(end);
  a.interval.n_barriers = static_cast<u32>(n_barriers);
  a.interval.bar_pos = d_bpos.p;
  a.interval.bar_dir = d_bdir.p;
  a.interval.nrows = 1;
  a.interval.ncols = 1;
  a.interval.bar_bucket = d_bucket.p;
  a.interval.bucket_shift = BAR_BUCKET_SHIFT;
  a.interval.n_buckets = static_cast<u32>(buckets.size());
  a.workspace = d_ws.p;
  a.image = d_img.p;
  a.mask = phase_mask;
  a.n = static_cast<u32>(n);
  std::memcpy(a.prng, prng, sizeof(a.prng));
  a.raws_out = h->d_phase_out.p;
  a.status_out = reinterpret_cast<u32*>(h->d_phase_out.p + 1);
  a.max_barriers = static_cast<u32>(n_barriers);
  {
    // the class the product would run this state in; WIDE also when a caller's move does not fit NARROW
    bool wide = modle_hip_size_class(&h->cfg, n) != 0;
    for (size_t i = 0; i < n; ++i) wide = wide || rev_moves[i] > 65533 || fwd_moves[i] > 65533;
    if (wide) test_phases_wide(a); else test_phases_narrow(a);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  uint64_t out[2] = {0, 0};
  HIP_TRY(hipMemcpy(out, h->d_phase_out.p, 16, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(image.data(), d_img.p, image.size() * 4, hipMemcpyDeviceToHost));
  modle_host::read_test_image(img, n, rev_pos, fwd_pos, epoch, rev_rank, fwd_rank, rev_moves,
                              fwd_moves, rev_coll, fwd_coll);
  if (raws_consumed != nullptr) *raws_consumed = out[0];
  if (static_cast<u32>(out[1]) != 0) {
    set_err(err, errlen, "device phase runner reported status " + std::to_string(out[1]));
    return MODLE_HIP_ERR_STATE;
  }
  return MODLE_HIP_OK;
}

int modle_hip_test_units(modle_hip_handle* h, uint32_t what, const uint64_t* in, size_t n,
                         uint64_t nrows, uint64_t ncols, uint32_t* contacts,
                         uint64_t* missed_updates, uint64_t* out, char* err, size_t errlen) {
  if (h == nullptr || n == 0 || in == nullptr) return MODLE_HIP_ERR_ARG;
  if (h->in_flight) return MODLE_HIP_ERR_STATE;
  if (n >= (1u << 24)) return MODLE_HIP_ERR_ARG;
  HIP_TRY(hipSetDevice(h->device));
  const auto layout = modle_host::workspace_layout(static_cast<u32>(n), 0, 4);
  DevBuf<char> d_ws;
  DevBuf<u64> d_in, d_out, d_missed;
  DevBuf<u32> d_contacts, d_status;
  const size_t nwords = what == UNIT_MATRIX_INCREMENT ? nrows * ncols + 1 : 1;
  if (what == UNIT_MATRIX_INCREMENT && (contacts == nullptr || nrows == 0 || ncols == 0)) {
    set_err(err, errlen, "matrix unit needs a contact buffer and its shape");
    return MODLE_HIP_ERR_ARG;
  }
  HIP_TRY(d_ws.ensure(layout.total_bytes));
  HIP_TRY(d_in.ensure(2 * n));
  HIP_TRY(d_out.ensure(2 * n));
  HIP_TRY(d_missed.ensure(1));
  HIP_TRY(d_status.ensure(1));
  HIP_TRY(d_contacts.ensure(nwords));
  HIP_TRY(hipMemcpy(d_in.p, in, 2 * n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(d_out.p, 0, 2 * n * 8));
  HIP_TRY(hipMemset(d_missed.p, 0, 8));
  HIP_TRY(hipMemset(d_status.p, 0xFF, 4));
  if (what == UNIT_MATRIX_INCREMENT) {
    HIP_TRY(hipMemcpy(d_contacts.p, contacts, nwords * 4, hipMemcpyHostToDevice));
    if (missed_updates != nullptr) HIP_TRY(hipMemcpy(d_missed.p, missed_updates, 8, hipMemcpyHostToDevice));
  }
  UnitArgs a;
  std::memset(&a, 0, sizeof(a));
  a.params = h->params;
  a.tables.jump = h->d_jump.p;
  a.tables.zig = h->d_zig.p;
  a.interval.start = 0;
  a.interval.end = 0xFFFFFFF0u;
  a.interval.contacts = d_contacts.p;
  a.interval.missed_updates = d_missed.p;
  a.interval.nrows = nrows == 0 ? 1 : nrows;
  a.interval.ncols = ncols == 0 ? 1 : ncols;
  a.workspace = d_ws.p;
  a.in = d_in.p;
  a.out = d_out.p;
  a.status_out = d_status.p;
  a.what = what;
  a.n = static_cast<u32>(n);
  if (modle_hip_size_class(&h->cfg, n) != 0) test_units_wide(a); else test_units_narrow(a);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  u32 st = 0;
  HIP_TRY(hipMemcpy(&st, d_status.p, 4, hipMemcpyDeviceToHost));
  if (st != 0) {
    set_err(err, errlen, "device unit runner reported status " + std::to_string(st));
    return MODLE_HIP_ERR_STATE;
  }
  if (out != nullptr) HIP_TRY(hipMemcpy(out, d_out.p, 2 * n * 8, hipMemcpyDeviceToHost));
  if (what == UNIT_MATRIX_INCREMENT) {
    HIP_TRY(hipMemcpy(contacts, d_contacts.p, nwords * 4, hipMemcpyDeviceToHost));
    if (missed_updates != nullptr) HIP_TRY(hipMemcpy(missed_updates, d_missed.p, 8, hipMemcpyDeviceToHost));
  }
  return MODLE_HIP_OK;
}

}  // extern "C"

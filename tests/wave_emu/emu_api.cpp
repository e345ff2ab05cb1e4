// emu_api.cpp -- runs the product's device code (modle_amd/csrc/sim_device.h) under the CPU lane
// emulator.  TEST INFRASTRUCTURE: lets the kernel logic be stepped against the oracle without a
// GPU.  It is not a fallback of the product (the product refuses to run without the HIP path).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "wave_emu.h"
// clang-format off
#include "sim_device.h"
// clang-format on
#include "host_prng.hpp"
#include "launch_common.hpp"
#include "zig_tables.h"

using namespace modle_dev;

#if defined(__has_feature)
#if __has_feature(memory_sanitizer)
#include <sanitizer/msan_interface.h>
#define EMU_MARK_UNINIT(ptr, bytes) __msan_allocated_memory((ptr), (bytes))
#endif
#endif
#ifndef EMU_MARK_UNINIT
#define EMU_MARK_UNINIT(ptr, bytes) ((void)0)
#endif

namespace {

struct LdsImage {
  std::vector<u64> ring;
  std::vector<u64> rng_state;
  std::vector<u64> rng_snap;
  std::vector<u64> jump;
  std::vector<u64> sort_lds;
  std::vector<u32> stage;
  WaveLds view() {
    WaveLds l;
    l.ring = ring.data();
    l.rng_state = rng_state.data();
    l.rng_snap = rng_snap.data();
    l.abort_flag = nullptr;
    l.mbox = nullptr;
    l.jump_table = jump.data();
    l.zig_norm_x = ZIG_NORM_X;
    l.zig_norm_y = ZIG_NORM_Y;
    l.zig_exp_x = ZIG_EXP_X;
    l.zig_exp_y = ZIG_EXP_Y;
    l.sort_lds = sort_lds.data();
    l.stage = stage.data();
    l.trace = nullptr;
    l.phase_ticks = nullptr;
    l.state_log = nullptr;
    l.state_log_cap = 0;
    l.trace_cap = 0;
    return l;
  }
  LdsImage()
      : ring(RNG_RING),
        rng_state(RNG_STATE_WORDS),
        rng_snap(8),
        jump(modle_host::build_jump_table(RNG_HOP)),
        sort_lds(SORT_LDS_CAP),
        stage(STAGE_CAP) {
    // on the device these scratch regions start with whatever the previous kernel left there
    EMU_MARK_UNINIT(ring.data(), ring.size() * 8);
    EMU_MARK_UNINIT(sort_lds.data(), sort_lds.size() * 8);
    EMU_MARK_UNINIT(stage.data(), stage.size() * 4);
  }
};

struct IntervalImage {
  std::vector<u32> bar_pos;
  std::vector<u8> bar_dir;
  std::vector<f64> stp_a, stp_i, occ;
  std::vector<u32> buckets;
  u64 missed = 0;
  Interval iv;
  IntervalImage(u64 start, u64 end, const u64* pos, const u8* dir, const f64* sa, const f64* si,
                size_t nb, u32* contacts, u64 nrows, u64 ncols, u64* occupancy) {
    for (size_t i = 0; i < nb; ++i) {
      bar_pos.push_back(static_cast<u32>(pos[i]));
      bar_dir.push_back(dir[i]);
      stp_a.push_back(sa ? sa[i] : 1.0);
      stp_i.push_back(si ? si[i] : 0.0);
      occ.push_back(modle_hip_occupancy_from_stp(stp_a.back(), stp_i.back()));
    }
    iv.start = static_cast<u32>(start);
    iv.end = static_cast<u32>(end);
    iv.n_barriers = static_cast<u32>(nb);
    iv.bar_pos = bar_pos.data();
    iv.bar_dir = bar_dir.data();
    iv.bar_stp_active = stp_a.data();
    iv.bar_stp_inactive = stp_i.data();
    iv.bar_occupancy = occ.data();
    iv.contacts = contacts;
    iv.occupancy_1d = occupancy;
    iv.missed_updates = &missed;
    iv.nrows = nrows;
    iv.ncols = ncols;
    buckets = modle_host::build_barrier_buckets(start, end, bar_pos);
    iv.bar_bucket = buckets.data();
    iv.bucket_shift = BAR_BUCKET_SHIFT;
    iv.n_buckets = static_cast<u32>(buckets.size());
  }
};

struct CellJob {
  const Params* p;
  const Interval* iv;
  const Task* task;
  Workspace ws;
  WaveLds lds;
  CellResult* res;
  u32 status;
};

void cell_body(void* arg) {
  CellJob* j = static_cast<CellJob*>(arg);
  CellResult r;
  const u32 st = simulate_cell(*j->p, *j->iv, *j->task, j->ws, j->lds, r);
  if (wave::lane() == 0) {
    *j->res = r;
    j->status = st;
  }
}

struct PhaseJob {
  const Params* p;
  const Interval* iv;
  Workspace ws;
  WaveLds lds;
  TestImage img;
  u32 mask, n;
  u64 prng[4];
  u64 raws;
  u32 status;
};

void phase_body(void* arg) {
  PhaseJob* j = static_cast<PhaseJob*>(arg);
  u64 raws = 0;
  const u32 st = run_test_phases(*j->p, *j->iv, j->ws, j->lds, j->img, j->mask, j->n, j->prng, raws);
  if (wave::lane() == 0) {
    j->raws = raws;
    j->status = st;
  }
}

}  // namespace

#ifdef MODLE_EMU_WRITE_TRACE
namespace emu_wtrace {
// the running cell's workspace, its last snapshot, and where the cell's readings stand: 0 = the cell's own
// first reading (sim_epoch.h: t_cell), then begin / end of a phase in turn
static const unsigned char* g_base = nullptr;
static size_t g_size = 0;
static std::vector<unsigned char> g_snap;
static int g_state = 0;
void start_cell(const void* base, size_t size) {
  g_base = static_cast<const unsigned char*>(base);
  g_size = size;
  g_snap.assign(size, 0);
  g_state = 0;
}
uint64_t tick() {
  if (g_state == 0) {
    g_state = 1;
    return 0;
  }
  if (g_state == 1) {  // a phase begins
    memcpy(g_snap.data(), g_base, g_size);
    g_state = 2;
    return 0;
  }
  uint64_t changed = 0;  // a phase ends: 32-bit words that differ, in bytes
  for (size_t i = 0; i + 4 <= g_size; i += 4) changed += memcmp(g_base + i, g_snap.data() + i, 4) != 0 ? 4 : 0;
  // ... and, in the high half of the reading, the aligned sectors of MODLE_EMU_WTRACE_SECTOR bytes that hold a
  // changed byte (round 5: what leaves an L2 that cannot keep a line from one phase to the next is whole
  // sectors, not the words that changed in them; profiles/r05*/write_accounting.txt)
#ifndef MODLE_EMU_WTRACE_SECTOR
#define MODLE_EMU_WTRACE_SECTOR 32
#endif
  uint64_t sectors = 0;
  const size_t skew = reinterpret_cast<uintptr_t>(g_base) % MODLE_EMU_WTRACE_SECTOR;  // (sectors of the ADDRESS)
  for (size_t i = 0; i < g_size;) {
    const size_t end = std::min(g_size, i + (MODLE_EMU_WTRACE_SECTOR - (i + skew) % MODLE_EMU_WTRACE_SECTOR));
    sectors += memcmp(g_base + i, g_snap.data() + i, end - i) != 0 ? 1 : 0;
    i = end;
  }
  g_state = 1;
  return changed | (sectors << 32);
}
}  // namespace emu_wtrace
static uint64_t g_phase_bytes[16];
#endif

extern "C" {

#ifdef MODLE_EMU_WRITE_TRACE
// bytes of device memory changed per phase (slot numbering of the PHASE macro), summed over the cells run so far
void emu_write_trace_read(uint64_t out[16], int reset) {
  memcpy(out, g_phase_bytes, sizeof(g_phase_bytes));
  if (reset) memset(g_phase_bytes, 0, sizeof(g_phase_bytes));
}
#endif

void emu_set_lane_schedule(unsigned schedule) { wave_emu::set_lane_schedule(schedule); }

// the device code's division by a wave-uniform bucket (sim_rng.h: udiv_by_uniform) against the
// plain quotient, over `n` (raw, range) pairs; returns the number of mismatches
uint64_t emu_check_udiv_by_uniform(const uint64_t* raws, const uint64_t* ranges, size_t n) {
  uint64_t bad = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint64_t bucket = modle_dev::uniform_int_bucket(ranges[i]);
    if (bucket > (uint64_t(1) << 62) || bucket < (uint64_t(1) << 24)) continue;
    const double inv = 1.0 / static_cast<double>(bucket);
    for (uint64_t raw : {raws[i], raws[i] / bucket * bucket, raws[i] / bucket * bucket + (bucket - 1),
                         uint64_t(0), ~uint64_t(0)}) {
      bad += modle_dev::udiv_by_uniform(raw, bucket, inv) != raw / bucket;
    }
  }
  return bad;
}

int emu_simulate_interval(const modle_hip_config* cfg, uint64_t start, uint64_t end,
                          const uint64_t* bar_pos, const uint8_t* bar_dir,
                          const double* bar_stp_active, const double* bar_stp_inactive,
                          size_t n_barriers, const modle_hip_task* tasks, size_t n_tasks,
                          uint32_t* contacts, uint64_t nrows, uint64_t ncols,
                          uint64_t* missed_updates, uint64_t* occupancy,
                          modle_hip_cell_result* results) {
  uint64_t max_lefs = 1;
  for (size_t t = 0; t < n_tasks; ++t) max_lefs = std::max<uint64_t>(max_lefs, tasks[t].num_lefs);
  if (const char* msg = modle_host::check_limits(*cfg, start, end, max_lefs, n_barriers)) {
    fprintf(stderr, "emu_simulate_interval: %s\n", msg);
    return MODLE_HIP_ERR_ARG;
  }
#ifndef MODLE_WIDE
  // (MODLE_EMU_FORCE_NARROW=1: tests/test_size_classes.py runs a set-up the rule classes WIDE through the NARROW
  // code on purpose, to see the net under the rule -- ERR_MOVE_RANGE -- catch it)
  if (modle_host::size_class_required(*cfg, max_lefs) != 0 && getenv("MODLE_EMU_FORCE_NARROW") == nullptr) {  // (tests/emu_sim.py loads libmodle_emu_wide.so for these)
    fprintf(stderr, "emu_simulate_interval: this set-up is of the WIDE size class, the build is NARROW\n");
    return MODLE_HIP_ERR_ARG;
  }
#endif
  const Params p = modle_host::make_params(*cfg);
  IntervalImage img(start, end, bar_pos, bar_dir, bar_stp_active, bar_stp_inactive, n_barriers,
                    contacts, nrows, ncols, occupancy);
  LdsImage lds;
  const auto layout = modle_host::workspace_layout(static_cast<u32>(max_lefs),
                                                   static_cast<u32>(n_barriers), p.hist_len);
  // device memory is not zero-initialised: poison the scratch so that reads of never-written
  // entries show up here as they would on the GPU
  std::vector<uint64_t> wsmem(layout.total_bytes / 8 + 1, 0xDEADBEEFCAFEF00Dull);
  EMU_MARK_UNINIT(wsmem.data(), wsmem.size() * 8);
  std::fill(lds.ring.begin(), lds.ring.end(), 0xDEADBEEFCAFEF00Dull);
  std::fill(lds.sort_lds.begin(), lds.sort_lds.end(), 0xDEADBEEFCAFEF00Dull);
  std::fill(lds.stage.begin(), lds.stage.end(), 0xDEADBEEFu);
  int rc = MODLE_HIP_OK;
  for (size_t t = 0; t < n_tasks; ++t) {
    Task task;
    task.interval = 0;
    task.num_lefs = static_cast<u32>(tasks[t].num_lefs);
    task.cell_id = tasks[t].cell_id;
    task.num_target_epochs = tasks[t].num_target_epochs;
    task.num_target_contacts = tasks[t].num_target_contacts;
    task.contacts_per_epoch = modle_hip_compute_contacts_per_epoch(cfg, tasks[t].num_lefs);
    memcpy(task.prng, tasks[t].prng, sizeof(task.prng));
    CellResult r;
    memset(&r, 0, sizeof(r));
    std::vector<u64> trace;
    WaveLds view = lds.view();
    const char* trace_path = t == 0 ? getenv("MODLE_EMU_TRACE") : nullptr;
    if (trace_path != nullptr) {
      trace.assign(static_cast<size_t>(4096) * TRACE_STAGES * TRACE_WORDS_PER_STAGE, 0);
      view.trace = trace.data();
      view.trace_cap = 4096;
    }
#ifdef MODLE_EMU_WRITE_TRACE
    view.phase_ticks = g_phase_bytes;
    emu_wtrace::start_cell(wsmem.data(), layout.total_bytes);
#endif
    CellJob job{&p, &img.iv, &task,
                modle_host::carve_workspace(wsmem.data(), static_cast<u32>(max_lefs),
                                            static_cast<u32>(n_barriers), p.hist_len),
                view, &r, 0};
    wave_emu::run_wave(cell_body, &job);
    if (trace_path != nullptr) {
      if (FILE* f = fopen(trace_path, "wb")) {
        fwrite(trace.data(), 8, trace.size(), f);
        fclose(f);
      }
    }
    if (job.status != 0) rc = MODLE_HIP_ERR_STATE;
    if (results != nullptr) memcpy(&results[t], &r, sizeof(r));
  }
  if (missed_updates != nullptr) *missed_updates += img.missed;
  return rc;
}

int emu_test_phases(const modle_hip_config* cfg, uint32_t phase_mask, uint64_t start,
                    uint64_t end, size_t n, uint64_t* rev_pos, uint64_t* fwd_pos, uint64_t* epoch,
                    uint64_t* rev_rank, uint64_t* fwd_rank, uint64_t* rev_moves,
                    uint64_t* fwd_moves, uint64_t* rev_coll, uint64_t* fwd_coll,
                    size_t n_barriers, const uint64_t* bar_pos, const uint8_t* bar_dir,
                    const uint8_t* bar_active, uint64_t prng[4], uint64_t* raws_consumed) {
  #ifndef MODLE_WIDE
  {
    bool wide = modle_host::size_class_required(*cfg, n) != 0;
    for (size_t i = 0; i < n; ++i) wide = wide || rev_moves[i] > 65533 || fwd_moves[i] > 65533;
    if (wide) {  // (tests/phase_backend.py loads libmodle_emu_wide.so for these)
      fprintf(stderr, "emu_test_phases: this state is of the WIDE size class, the build is NARROW\n");
      return MODLE_HIP_ERR_ARG;
    }
  }
#endif
  const Params p = modle_host::make_params(*cfg);
  IntervalImage img_iv(start, end, bar_pos, bar_dir, nullptr, nullptr, n_barriers, nullptr, 1, 1,
                       nullptr);
  LdsImage lds;
  const auto layout =
      modle_host::workspace_layout(static_cast<u32>(n), static_cast<u32>(n_barriers), 4);
  std::vector<uint64_t> wsmem(layout.total_bytes / 8 + 1);
  Workspace ws = modle_host::carve_workspace(wsmem.data(), static_cast<u32>(n),
                                             static_cast<u32>(n_barriers), 4);
  std::vector<u32> image(9 * n);
  TestImage img;
  modle_host::fill_test_image(image.data(), n, rev_pos, fwd_pos, epoch, rev_rank, fwd_rank,
                              rev_moves, fwd_moves, rev_coll, fwd_coll, img);
  for (size_t i = 0; i < n_barriers; ++i) ws.bar_active[i] = bar_active[i];
  PhaseJob job{&p, &img_iv.iv, ws, lds.view(), img, phase_mask, static_cast<u32>(n),
               {0, 0, 0, 0}, 0, 0};
  memcpy(job.prng, prng, sizeof(job.prng));
  wave_emu::run_wave(phase_body, &job);
  modle_host::read_test_image(img, n, rev_pos, fwd_pos, epoch, rev_rank, fwd_rank, rev_moves,
                              fwd_moves, rev_coll, fwd_coll);
  if (raws_consumed != nullptr) *raws_consumed = job.raws;
  return job.status == 0 ? MODLE_HIP_OK : MODLE_HIP_ERR_STATE;
}

struct UnitJob {
  const Params* p;
  const Interval* iv;
  Workspace ws;
  WaveLds lds;
  u32 what, n;
  const u64* in;
  u64* out;
  u32 status;
};
static void unit_body(void* arg) {
  UnitJob* j = static_cast<UnitJob*>(arg);
  const u32 st = run_test_units(*j->p, *j->iv, j->ws, j->lds, j->what, j->in, j->n, j->out);
  if (wave::lane() == 0) j->status = st;
}

int emu_test_units(const modle_hip_config* cfg, uint32_t what, const uint64_t* in, size_t n,
                   uint64_t nrows, uint64_t ncols, uint32_t* contacts, uint64_t* missed_updates,
                   uint64_t* out) {
#ifndef MODLE_WIDE
  if (modle_host::size_class_required(*cfg, n) != 0) {  // (the tests load libmodle_emu_wide.so for these)
    fprintf(stderr, "emu_test_units: %zu elements are of the WIDE size class, the build is NARROW\n", n);
    return MODLE_HIP_ERR_ARG;
  }
#endif
  const Params p = modle_host::make_params(*cfg);
  uint32_t dummy = 0;
  IntervalImage img(0, 0xFFFFFFF0ull, nullptr, nullptr, nullptr, nullptr, 0,
                    contacts != nullptr ? contacts : &dummy, nrows == 0 ? 1 : nrows,
                    ncols == 0 ? 1 : ncols, nullptr);
  if (missed_updates != nullptr) img.missed = *missed_updates;
  LdsImage lds;
  const auto layout = modle_host::workspace_layout(static_cast<u32>(n), 0, 4);
  std::vector<uint64_t> wsmem(layout.total_bytes / 8 + 1);
  UnitJob job{&p, &img.iv, modle_host::carve_workspace(wsmem.data(), static_cast<u32>(n), 0, 4),
              lds.view(), what, static_cast<u32>(n), in, out, 0};
  wave_emu::run_wave(unit_body, &job);
  if (missed_updates != nullptr) *missed_updates = img.missed;
  return job.status == 0 ? MODLE_HIP_OK : MODLE_HIP_ERR_STATE;
}

}  // extern "C"

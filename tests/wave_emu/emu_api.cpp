// emu_api.cpp -- runs the product's device code (modle_amd/csrc/sim_device.h) under the CPU lane
// emulator.  TEST INFRASTRUCTURE: lets the kernel logic be stepped against the oracle without a
// GPU.  It is not a fallback of the product (the product refuses to run without the HIP path).
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

namespace {

struct LdsImage {
  std::vector<u64> ring;
  std::vector<u64> jump;
  std::vector<u32> list;
  WaveLds view() {
    WaveLds l;
    l.ring = ring.data();
    l.jump_table = jump.data();
    l.zig_norm_x = ZIG_NORM_X;
    l.zig_norm_y = ZIG_NORM_Y;
    l.zig_exp_x = ZIG_EXP_X;
    l.zig_exp_y = ZIG_EXP_Y;
    l.list = list.data();
    return l;
  }
  LdsImage() : ring(RNG_RING), jump(modle_host::build_jump_table(RNG_BLOCK)), list(LIST_CAP) {}
};

struct IntervalImage {
  std::vector<u32> bar_pos;
  std::vector<u8> bar_dir;
  std::vector<f64> stp_a, stp_i, occ;
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
  u32 mask, n;
  u64 prng[4];
  u64 raws;
  u32 status;
};

void phase_body(void* arg) {
  PhaseJob* j = static_cast<PhaseJob*>(arg);
  u64 raws = 0;
  const u32 st = run_test_phases(*j->p, *j->iv, j->ws, j->lds, j->mask, j->n, j->prng, raws);
  if (wave::lane() == 0) {
    j->raws = raws;
    j->status = st;
  }
}

}  // namespace

extern "C" {

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
  const Params p = modle_host::make_params(*cfg);
  IntervalImage img(start, end, bar_pos, bar_dir, bar_stp_active, bar_stp_inactive, n_barriers,
                    contacts, nrows, ncols, occupancy);
  LdsImage lds;
  const auto layout = modle_host::workspace_layout(static_cast<u32>(max_lefs),
                                                   static_cast<u32>(n_barriers), p.hist_len);
  std::vector<uint64_t> wsmem(layout.total_bytes / 8 + 1);
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
    CellJob job{&p, &img.iv, &task,
                modle_host::carve_workspace(wsmem.data(), static_cast<u32>(max_lefs),
                                            static_cast<u32>(n_barriers), p.hist_len),
                lds.view(), &r, 0};
    wave_emu::run_wave(cell_body, &job);
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
  const Params p = modle_host::make_params(*cfg);
  IntervalImage img(start, end, bar_pos, bar_dir, nullptr, nullptr, n_barriers, nullptr, 1, 1,
                    nullptr);
  LdsImage lds;
  const auto layout =
      modle_host::workspace_layout(static_cast<u32>(n), static_cast<u32>(n_barriers), 4);
  std::vector<uint64_t> wsmem(layout.total_bytes / 8 + 1);
  Workspace ws = modle_host::carve_workspace(wsmem.data(), static_cast<u32>(n),
                                             static_cast<u32>(n_barriers), 4);
  for (size_t i = 0; i < n; ++i) {
    ws.rev_pos[i] = modle_host::pos_to_dev(rev_pos[i]);
    ws.fwd_pos[i] = modle_host::pos_to_dev(fwd_pos[i]);
    ws.epoch[i] = modle_host::pos_to_dev(epoch[i]);
    ws.rev_rank[i] = static_cast<u32>(rev_rank[i]);
    ws.fwd_rank[i] = static_cast<u32>(fwd_rank[i]);
    ws.rev_moves[i] = static_cast<u32>(rev_moves[i]);
    ws.fwd_moves[i] = static_cast<u32>(fwd_moves[i]);
    ws.rev_coll[i] = modle_host::coll_to_dev(rev_coll[i]);
    ws.fwd_coll[i] = modle_host::coll_to_dev(fwd_coll[i]);
  }
  for (size_t i = 0; i < n_barriers; ++i) ws.bar_active[i] = bar_active[i];
  PhaseJob job{&p, &img.iv, ws, lds.view(), phase_mask, static_cast<u32>(n), {0, 0, 0, 0}, 0, 0};
  memcpy(job.prng, prng, sizeof(job.prng));
  wave_emu::run_wave(phase_body, &job);
  for (size_t i = 0; i < n; ++i) {
    rev_pos[i] = modle_host::pos_to_abi(ws.rev_pos[i]);
    fwd_pos[i] = modle_host::pos_to_abi(ws.fwd_pos[i]);
    epoch[i] = modle_host::pos_to_abi(ws.epoch[i]);
    rev_rank[i] = ws.rev_rank[i];
    fwd_rank[i] = ws.fwd_rank[i];
    rev_moves[i] = ws.rev_moves[i];
    fwd_moves[i] = ws.fwd_moves[i];
    rev_coll[i] = modle_host::coll_to_abi(ws.rev_coll[i]);
    fwd_coll[i] = modle_host::coll_to_abi(ws.fwd_coll[i]);
  }
  if (raws_consumed != nullptr) *raws_consumed = job.raws;
  return job.status == 0 ? MODLE_HIP_OK : MODLE_HIP_ERR_STATE;
}

}  // extern "C"

// MemorySanitizer driver for the device code under the lane emulator (test infrastructure):
// runs whole cells so that any read of uninitialised registers / LDS / workspace is reported.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "modle_hip.h"

extern "C" int emu_simulate_interval(const modle_hip_config* cfg, uint64_t start, uint64_t end,
                                     const uint64_t* bar_pos, const uint8_t* bar_dir,
                                     const double* bar_stp_active, const double* bar_stp_inactive,
                                     size_t n_barriers, const modle_hip_task* tasks, size_t n_tasks,
                                     uint32_t* contacts, uint64_t nrows, uint64_t ncols,
                                     uint64_t* missed_updates, uint64_t* occupancy,
                                     modle_hip_cell_result* results);

int main(int argc, char** argv) {
  const uint64_t size = argc > 1 ? strtoull(argv[1], nullptr, 10) : 5000000;
  const bool with_barriers = argc > 2 && atoi(argv[2]) != 0;
  const size_t n_cells = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
  modle_hip_config cfg;
  modle_hip_config_default(&cfg);
  cfg.num_cells = 8;
  cfg.simulate_chromosomes_wo_barriers = 1;
  if (!with_barriers) cfg.number_of_lefs_per_mbp = 16.0;
  // optional: LEFs per Mb, processivity, skip the burn-in, target contact density (the regimes of the
  // rank update that borrows the generator's ring, of BASELINE configs[4], ...)
  if (argc > 4 && atof(argv[4]) > 0) cfg.number_of_lefs_per_mbp = atof(argv[4]);
  if (argc > 5 && atof(argv[5]) > 0) cfg.avg_lef_processivity = static_cast<uint64_t>(atof(argv[5]));
  if (argc > 6) cfg.skip_burnin = atoi(argv[6]) != 0;
  if (argc > 7 && atof(argv[7]) > 0) cfg.target_contact_density = atof(argv[7]);
  if (argc > 8 && atof(argv[8]) > 0) {
    cfg.lef_bar_minor_collision_pblock = atof(argv[8]);
    cfg.soft_stall_lef_stability_multiplier = 2.0;
  }
  if (argc > 9 && atoi(argv[9]) > 0) cfg.max_burnin_epochs = static_cast<uint64_t>(atoi(argv[9]));
  // named settings anywhere behind the positional ones: spacing=<bp> (a barrier every so many bp instead of
  // one per 30-120 kb), major=<p> / minor=<p> (blocking probabilities: fractional ones make LEF-BAR detection
  // draw Bernoulli trials), softstall=<x>
  uint64_t spacing = 0;
  for (int i = 1; i < argc; ++i) {
    if (strncmp(argv[i], "spacing=", 8) == 0) spacing = strtoull(argv[i] + 8, nullptr, 10);
    if (strncmp(argv[i], "major=", 6) == 0) cfg.lef_bar_major_collision_pblock = atof(argv[i] + 6);
    if (strncmp(argv[i], "minor=", 6) == 0) cfg.lef_bar_minor_collision_pblock = atof(argv[i] + 6);
    if (strncmp(argv[i], "softstall=", 10) == 0) cfg.soft_stall_lef_stability_multiplier = atof(argv[i] + 10);
  }
  char err[256];
  if (modle_hip_config_transform(&cfg, err, sizeof(err)) < 0) return 2;
  std::vector<uint64_t> bp;
  std::vector<uint8_t> bd;
  std::vector<double> sa, si;
  if (with_barriers) {
    uint64_t x = 12345;
    const uint64_t lo = spacing != 0 ? spacing / 2 + 1 : 30000, span = spacing != 0 ? spacing : 90000;
    for (uint64_t pos = spacing != 0 ? spacing : 20000; pos + 20000 < size;
         pos += lo + (x = x * 6364136223846793005ull + 1442695040888963407ull) % span) {
      bp.push_back(pos);
      bd.push_back(((x >> 40) & 1) ? MODLE_HIP_DIR_FWD : MODLE_HIP_DIR_REV);
      sa.push_back(modle_hip_stp_active_from_occupancy(cfg.barrier_not_occupied_stp, 0.6 + 0.3 * ((x >> 20) % 100) / 100.0));
      si.push_back(cfg.barrier_not_occupied_stp);
    }
  }
  std::vector<modle_hip_task> tasks(cfg.num_cells);
  modle_hip_make_tasks(&cfg, "chrT", size, 0, size, 0, tasks.data());
  uint64_t nr = 0, nc = 0;
  modle_hip_matrix_shape(&cfg, size, &nr, &nc);
  std::vector<uint32_t> contacts(nr * nc + 1, 0);
  std::vector<uint64_t> occ(nc, 0);
  uint64_t missed = 0;
  std::vector<modle_hip_cell_result> res(n_cells);
  const int rc = emu_simulate_interval(&cfg, 0, size, bp.data(), bd.data(), sa.data(), si.data(), bp.size(),
                                       tasks.data(), n_cells, contacts.data(), nr, nc, &missed, occ.data(), res.data());
  printf("rc=%d epochs=%llu contacts=%llu barriers=%zu\n", rc, (unsigned long long)res[0].epochs,
         (unsigned long long)res[0].num_contacts, bp.size());
  return rc;
}

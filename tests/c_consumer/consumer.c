/* A plain C11 consumer of include/modle_hip.h and include/modle_cooler.h: compiled with
 * -std=c11 -Wall -Wextra -Werror by tests/test_host_logic.py, linked against the two shared
 * libraries, and run on the host-logic entry points (no GPU needed).  What it prints is compared
 * with the Python view of the same calls. */
#include <inttypes.h>
#include <stdio.h>
#include <string.h>

#include "modle_bigwig.h"
#include "modle_cooler.h"
#include "modle_genome.h"
#include "modle_hip.h"

int main(int argc, char** argv) {
  char err[256] = {0};
  modle_hip_config cfg;
  modle_hip_config_default(&cfg);
  cfg.num_cells = 4;
  if (modle_hip_config_transform(&cfg, err, sizeof(err)) != MODLE_HIP_OK) {
    fprintf(stderr, "transform: %s\n", err);
    return 1;
  }
  const uint64_t size = 5000000;
  uint64_t nrows = 0, ncols = 0;
  modle_hip_matrix_shape(&cfg, size, &nrows, &ncols);
  modle_hip_task tasks[4];
  if (modle_hip_make_tasks(&cfg, "chrC", size, 0, size, 0, tasks) != MODLE_HIP_OK) return 2;
  printf("hash %" PRIu64 "\n", modle_hip_interval_hash("chrC", size, 0, size, cfg.seed));
  printf("shape %" PRIu64 " %" PRIu64 "\n", nrows, ncols);
  printf("nlefs %" PRIu64 "\n", modle_hip_compute_num_lefs(&cfg, size));
  for (int i = 0; i < 4; ++i)
    printf("task %" PRIu64 " %" PRIu64 " %" PRIu64 " %" PRIu64 "\n", tasks[i].cell_id,
           tasks[i].num_target_contacts, tasks[i].prng[0], tasks[i].prng[3]);
  /* barriers in BED order: the library sorts (ExtrusionBarriers::sort) */
  uint64_t pos[3] = {900, 100, 500};
  uint8_t dir[3] = {MODLE_HIP_DIR_FWD, MODLE_HIP_DIR_REV, MODLE_HIP_DIR_FWD};
  double sa[3] = {0.9, 0.8, 0.7}, si[3] = {0.1, 0.2, 0.3};
  modle_hip_sort_barriers(pos, dir, sa, si, 3);
  printf("sorted %" PRIu64 " %" PRIu64 " %" PRIu64 " %u %.1f %.1f\n", pos[0], pos[1], pos[2],
         (unsigned)dir[0], sa[0], si[2]);
  printf("stp %.17g\n", modle_hip_stp_active_from_occupancy(cfg.barrier_not_occupied_stp, 0.85));
  /* without a GPU the device path must refuse loudly, never fall back */
  modle_hip_handle* h = modle_hip_create(&cfg, 0, err, sizeof(err));
  printf("create %s\n", h != NULL ? "ok" : "refused");
  if (h != NULL) modle_hip_destroy(h);
  {
    /* genome import: one chromosome, one barrier */
    const char* cs = "chrC\t50000\n";
    const char* bed = "chrC\t100\t120\tb\t0.9\t+\n";
    modle_genome* g = NULL;
    if (modle_genome_import(cs, strlen(cs), bed, strlen(bed), NULL, 0, &cfg, 0, &g, err,
                            sizeof(err)) != MODLE_GENOME_OK) {
      fprintf(stderr, "genome: %s\n", err);
      return 6;
    }
    modle_genome_interval info;
    uint64_t bpos = 0;
    uint8_t bdir = 0;
    double bsa = 0, bsi = 0;
    if (modle_genome_interval_info(g, 0, &info) != MODLE_GENOME_OK ||
        modle_genome_interval_barriers(g, 0, &bpos, &bdir, &bsa, &bsi) != MODLE_GENOME_OK)
      return 7;
    printf("genome %zu %" PRIu64 " %" PRIu64 " %u\n", modle_genome_num_chromosomes(g), info.end,
           bpos, (unsigned)bdir);
    modle_genome_free(g);
  }
  if (argc > 1) {
    const char* names[1] = {"chrC"};
    const uint32_t sizes[1] = {50000};
    modle_cool_file* f = NULL;
    if (modle_cool_create(argv[1], 1, names, sizes, 1, 10000, "asm", "c-consumer", NULL, &f, err,
                          sizeof(err)) != MODLE_COOL_OK) {
      fprintf(stderr, "cooler: %s\n", err);
      return 3;
    }
    uint32_t band[2 * 5 + 1];
    memset(band, 0, sizeof(band));
    band[0] = 7; /* (0,0) */
    band[3] = 2; /* col 1, diagonal 1: (0,1) */
    if (modle_cool_append_matrix(f, 0, 0, band, 2, 5, err, sizeof(err)) != MODLE_COOL_OK) return 4;
    if (modle_cool_close(f, err, sizeof(err)) != MODLE_COOL_OK) return 5;
    printf("cooler ok\n");
    if (argc > 2) {
      modle_bw_file* bw = NULL;
      const uint64_t occ[5] = {1, 4, 0, 2, 4};
      if (modle_bw_create(argv[2], 1, names, sizes, 1, &bw, err, sizeof(err)) != MODLE_BW_OK) return 8;
      if (modle_bw_write_occupancy(bw, 0, occ, 5, 10000, 0, err, sizeof(err)) != MODLE_BW_OK) return 9;
      if (modle_bw_close(bw, err, sizeof(err)) != MODLE_BW_OK) return 10;
      printf("bigwig ok\n");
    }
  }
  return 0;
}

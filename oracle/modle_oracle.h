/* modle_oracle.h -- CPU restatement ("oracle") of MoDLE's Simulation::simulate_one_cell path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (modle_amd/, include/) may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * there only as the checker / the timed CPU baseline.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 *
 * PARITY STATUS
 *   - collision / move / ranking logic: pinned by the reference's own deterministic unit-test
 *     vectors (tests/golden/reference_kats.json, extracted from
 *     test/units/simulation_cpu/simulation_{simple,complex}_unit_test.cpp).
 *   - PRNG (xoshiro256++ / SplitMix64, xoshiro-cpp 1.1) and bernoulli convention: pinned by
 *     the reference's seed-dependent tests "Simulation 011/012" and by the values quoted in
 *     SURVEY.md section 8c; XXH3 seeding pinned against the python `xxhash` module.
 *   - Boost.Random 1.88 normal / poisson / binomial / uniform_int / generate_canonical and
 *     cpp-sort tie order: the sources are NOT in /root/reference nor in this image.  They are
 *     restated from the published algorithms (Marsaglia-Tsang ziggurat, Hoermann PTRD / BTRD,
 *     bucket rejection).  **parity unpinned** for those draws against an official binary.
 */
#ifndef MODLE_ORACLE_H
#define MODLE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MO_UNBOUND UINT64_MAX

/* Collision word: idx | event << 56  (collision_encoding.hpp:54-109; uint_fast32_t is 64-bit on
 * Linux/glibc). */
#define MO_EV_COLLISION 0x10u
#define MO_EV_CHROM_BOUNDARY 0x08u
#define MO_EV_LEF_BAR 0x04u
#define MO_EV_LEF_LEF_PRIMARY 0x02u
#define MO_EV_LEF_LEF_SECONDARY 0x01u
#define MO_EVENT_SHIFT 56
#define MO_INDEX_MASK ((UINT64_C(1) << 55) - 1)

/* contact_sampling_strategy flags (simulation_config.hpp:33-38; numeric values are internal) */
#define MO_CS_NOISIFY 1u
#define MO_CS_TAD 2u
#define MO_CS_LOOP 4u

/* barrier blocking direction (dna.hpp; values internal): */
#define MO_DIR_FWD 1u
#define MO_DIR_REV 2u

typedef struct mo_prng {
  uint64_t s[4];
  uint64_t count; /* raw 64-bit outputs drawn so far (bookkeeping only) */
} mo_prng_t;

/* Post-`transform_args` Config fields read by the path (simulation_config.hpp:47-113,
 * cli.cpp:886-1016).  All members are 8 bytes wide so the layout is trivially mirrored from
 * ctypes; the product's `modle_hip_params` has the same member order. */
typedef struct mo_params {
  uint64_t bin_size;
  uint64_t diagonal_width;
  uint64_t rev_extrusion_speed;
  uint64_t fwd_extrusion_speed;
  double rev_extrusion_speed_std; /* absolute (bp) */
  double fwd_extrusion_speed_std;
  uint64_t rev_extrusion_speed_burnin;
  uint64_t fwd_extrusion_speed_burnin;
  double prob_of_lef_release;
  double prob_of_lef_release_burnin;
  double hard_stall_lef_stability_multiplier;
  double soft_stall_lef_stability_multiplier;
  double probability_of_extrusion_unit_bypass;
  double lef_bar_major_collision_pblock;
  double lef_bar_minor_collision_pblock;
  uint64_t contact_sampling_interval;
  uint64_t contact_sampling_strategy;
  double tad_to_loop_contact_ratio;
  double genextreme_mu;
  double genextreme_sigma;
  double genextreme_xi;
  double target_contact_density; /* < 0 => stop on epochs */
  uint64_t target_simulation_epochs;
  uint64_t skip_burnin;
  uint64_t burnin_history_length;
  uint64_t burnin_smoothing_window_size;
  uint64_t min_burnin_epochs;
  uint64_t max_burnin_epochs;
  uint64_t burnin_target_epochs_for_lef_activation;
  uint64_t track_1d_lef_position;
  double number_of_lefs_per_mbp;
  uint64_t num_cells;
  uint64_t seed;
  uint64_t simulate_chromosomes_wo_barriers;
  /* raw CLI-level inputs (unused by the oracle's simulation code; present so that the struct
   * layout equals the product's modle_hip_config and one ctypes class serves both) */
  uint64_t avg_lef_processivity;
  double burnin_speed_coefficient;
  double extrusion_barrier_occupancy;
  double barrier_occupied_stp;
  double barrier_not_occupied_stp;
  uint64_t probability_normalization_factor;
  uint64_t normalize_probabilities;
  uint64_t rev_extrusion_speed_set;
  uint64_t fwd_extrusion_speed_set;
  uint64_t extrusion_barrier_occupancy_set;
} mo_params_t;

typedef struct mo_task {
  uint64_t id;
  uint64_t cell_id;
  uint64_t num_target_epochs;
  uint64_t num_target_contacts;
  uint64_t num_lefs;
  uint64_t prng[4];
} mo_task_t;

typedef struct mo_cell_result {
  uint64_t epochs;         /* State::epoch at exit */
  uint64_t burnin_epochs;  /* State::num_burnin_epochs */
  uint64_t num_contacts;   /* State::num_contacts */
  uint64_t raws_consumed;  /* number of 64-bit PRNG outputs drawn by the cell */
  uint64_t prng_final[4];  /* PRNG state at exit */
  uint64_t sum_active_lefs; /* sum over executed epochs of num_active_lefs (roofline bytes) */
  uint64_t sampling_events; /* total contact sampling events executed (roofline bytes) */
  uint64_t sim_epochs;      /* epochs whose move/collision phase ran */
} mo_cell_result_t;

/* ---- PRNG ----------------------------------------------------------------------------- */
void mo_prng_seed(mo_prng_t* g, uint64_t seed);
uint64_t mo_prng_next(mo_prng_t* g);
void mo_prng_jump(mo_prng_t* g);

/* ---- distributions (Boost.Random 1.88 restated; see header note) -------------------------- */
int mo_bernoulli(mo_prng_t* g, double p);
double mo_canonical(mo_prng_t* g);
double mo_uniform_01(mo_prng_t* g);
uint64_t mo_uniform_int(mo_prng_t* g, uint64_t lo, uint64_t hi);
double mo_normal(mo_prng_t* g, double mean, double sigma);
uint64_t mo_poisson(mo_prng_t* g, double mean);
int64_t mo_binomial(mo_prng_t* g, int64_t t, double p);
double mo_genextreme(mo_prng_t* g, double mu, double sigma, double xi);

/* ---- hashing / task derivation ------------------------------------------------------------- */
uint64_t mo_xxh3_64(const void* data, size_t len, uint64_t seed);
uint64_t mo_interval_hash(const char* chrom_name, uint64_t chrom_size, uint64_t start,
                          uint64_t end, uint64_t seed);
uint64_t mo_compute_num_lefs(const mo_params_t* p, uint64_t size_bp);
uint64_t mo_compute_contacts_per_epoch(const mo_params_t* p, uint64_t nlefs);
void mo_matrix_shape(const mo_params_t* p, uint64_t size_bp, uint64_t* nrows, uint64_t* ncols);
/* fills tasks[0..num_cells) for one interval the way run_simulate does */
void mo_make_tasks(const mo_params_t* p, const char* chrom_name, uint64_t chrom_size,
                   uint64_t start, uint64_t end, uint64_t first_task_id, mo_task_t* tasks);

/* ---- barrier maths -------------------------------------------------------------------------- */
double mo_stp_active_from_occupancy(double stp_inactive, double occupancy);
double mo_occupancy_from_stp(double stp_active, double stp_inactive);

/* ---- phase-level entry points (mirror Simulation::test_* hooks) ------------------------- */
void mo_rank_lefs(size_t n, const uint64_t* rev_pos, const uint64_t* fwd_pos,
                  const uint64_t* epoch, uint64_t* rev_rank, uint64_t* fwd_rank, int init_buffers);
void mo_adjust_moves(uint64_t start, uint64_t end, size_t n, const uint64_t* rev_pos,
                     const uint64_t* fwd_pos, const uint64_t* epoch, const uint64_t* rev_rank,
                     const uint64_t* fwd_rank, uint64_t* rev_moves, uint64_t* fwd_moves);
void mo_clamp_moves(uint64_t start, uint64_t end, size_t n, const uint64_t* rev_pos,
                    const uint64_t* fwd_pos, const uint64_t* epoch, uint64_t* rev_moves,
                    uint64_t* fwd_moves);
void mo_detect_units_at_interval_boundaries(uint64_t start, uint64_t end, size_t n,
                                            const uint64_t* rev_pos, const uint64_t* fwd_pos,
                                            const uint64_t* epoch, const uint64_t* rev_rank,
                                            const uint64_t* fwd_rank, const uint64_t* rev_moves,
                                            const uint64_t* fwd_moves, uint64_t* rev_coll,
                                            uint64_t* fwd_coll, uint64_t* n5, uint64_t* n3);
void mo_detect_lef_bar_collisions(const mo_params_t* p, size_t n, const uint64_t* rev_pos,
                                  const uint64_t* fwd_pos, const uint64_t* epoch,
                                  const uint64_t* rev_rank, const uint64_t* fwd_rank,
                                  const uint64_t* rev_moves, const uint64_t* fwd_moves,
                                  size_t nb, const uint64_t* bar_pos, const uint8_t* bar_dir,
                                  const uint8_t* bar_active, uint64_t* rev_coll,
                                  uint64_t* fwd_coll, mo_prng_t* g, uint64_t n5, uint64_t n3);
void mo_detect_primary_lef_lef_collisions(const mo_params_t* p, size_t n, const uint64_t* rev_pos,
                                          const uint64_t* fwd_pos, const uint64_t* rev_rank,
                                          const uint64_t* fwd_rank, const uint64_t* rev_moves,
                                          const uint64_t* fwd_moves, size_t nb,
                                          const uint64_t* bar_pos, uint64_t* rev_coll,
                                          uint64_t* fwd_coll, mo_prng_t* g, uint64_t n5,
                                          uint64_t n3);
void mo_correct_moves_for_lef_bar_collisions(size_t n, const uint64_t* rev_pos,
                                             const uint64_t* fwd_pos, const uint64_t* bar_pos,
                                             uint64_t* rev_moves, uint64_t* fwd_moves,
                                             const uint64_t* rev_coll, const uint64_t* fwd_coll);
void mo_correct_moves_for_primary_lef_lef_collisions(size_t n, const uint64_t* rev_pos,
                                                     const uint64_t* fwd_pos,
                                                     const uint64_t* rev_rank,
                                                     const uint64_t* fwd_rank, uint64_t* rev_moves,
                                                     uint64_t* fwd_moves, const uint64_t* rev_coll,
                                                     const uint64_t* fwd_coll);
void mo_process_secondary_lef_lef_collisions(const mo_params_t* p, size_t n,
                                             const uint64_t* rev_pos, const uint64_t* fwd_pos,
                                             const uint64_t* rev_rank, const uint64_t* fwd_rank,
                                             uint64_t* rev_moves, uint64_t* fwd_moves,
                                             uint64_t* rev_coll, uint64_t* fwd_coll, mo_prng_t* g,
                                             uint64_t n5, uint64_t n3);
void mo_fix_secondary_lef_lef_collisions(uint64_t start, uint64_t end, size_t n, uint64_t* rev_pos,
                                         uint64_t* fwd_pos, uint64_t* rev_rank, uint64_t* fwd_rank,
                                         uint64_t* rev_moves, uint64_t* fwd_moves,
                                         uint64_t* rev_coll, uint64_t* fwd_coll, uint64_t n5,
                                         uint64_t n3);
/* process_collisions = the seven calls above in the reference's order (simulation.cpp:763-793);
 * with_fix = 0 reproduces Simulation::test_process_collisions (simulation.hpp:499-528). */
void mo_process_collisions(const mo_params_t* p, uint64_t start, uint64_t end, size_t n,
                           uint64_t* rev_pos, uint64_t* fwd_pos, const uint64_t* epoch,
                           uint64_t* rev_rank, uint64_t* fwd_rank, uint64_t* rev_moves,
                           uint64_t* fwd_moves, size_t nb, const uint64_t* bar_pos,
                           const uint8_t* bar_dir, const uint8_t* bar_active, uint64_t* rev_coll,
                           uint64_t* fwd_coll, mo_prng_t* g, int with_fix);
void mo_generate_moves(const mo_params_t* p, uint64_t start, uint64_t end, size_t n,
                       const uint64_t* rev_pos, const uint64_t* fwd_pos, const uint64_t* epoch,
                       const uint64_t* rev_rank, const uint64_t* fwd_rank, uint64_t* rev_moves,
                       uint64_t* fwd_moves, int burnin_completed, mo_prng_t* g, int adjust);

/* ---- whole-cell simulation --------------------------------------------------------------- */
/* contacts: nrows*ncols+1 uint32 (band layout), accumulated into with atomic adds;
 * missed: updates that fell outside the band; occupancy: ncols uint64 or NULL. */
/* generator policy of the in-cell stream: 0 = xoshiro256++ (reference), 1 = PHILOX (see .c) */
void mo_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);
void mo_set_rng_policy(int philox);
int mo_get_rng_policy(void);
/* software log / exp / pow shared with the device code (modle_amd/csrc/modle_math.h) */
double mo_math_log(double x);
double mo_math_exp(double x);
double mo_math_pow(double x, double y);
/* unit hooks for the reference's stats / contact-matrix / collision-encoding tests */
void mo_loop_size_stats(size_t n, const uint64_t* rev_pos, const uint64_t* fwd_pos, double* avg,
                        double* ssd, double* var, double* std);
void mo_matrix_increment(uint32_t* contacts, uint64_t nrows, uint64_t ncols, uint64_t row,
                         uint64_t col, uint64_t* missed);
uint64_t mo_collision_word(uint64_t idx, unsigned ev);
unsigned mo_collision_predicates(uint64_t word);
/* Simulation::select_and_bind_lefs (simulation.cpp:988-993); scratch: 3 n words */
void mo_select_and_bind_lefs(uint64_t start, uint64_t end, size_t n, uint64_t* rev_pos,
                             uint64_t* fwd_pos, uint64_t* epoch, uint64_t* rev_rank,
                             uint64_t* fwd_rank, uint64_t epoch_now, mo_prng_t* g,
                             uint64_t* scratch);
/* one cell with its model-internal-state log (Simulation::dump_stats): 10 words per record in
 * the layout of MODLE_HIP_STATE_LOG_WORDS; returns the number of records written */
size_t mo_simulate_cell_with_state_log(const mo_params_t* p, uint64_t start, uint64_t end, size_t nb,
                                       const uint64_t* bar_pos, const uint8_t* bar_dir,
                                       const double* bar_stp_active, const double* bar_stp_inactive,
                                       const mo_task_t* task, uint32_t* contacts, uint64_t nrows,
                                       uint64_t ncols, uint64_t* missed, uint64_t* occupancy,
                                       mo_cell_result_t* res, uint64_t* log, size_t cap);
/* ExtrusionBarriers::sort (extrusion_barriers.cpp:237-257), in place */
void mo_sort_barriers(size_t nb, uint64_t* pos, uint8_t* dir, double* stp_active,
                      double* stp_inactive);
int mo_simulate_cell(const mo_params_t* p, uint64_t start, uint64_t end, size_t nb,
                     const uint64_t* bar_pos, const uint8_t* bar_dir, const double* bar_stp_active,
                     const double* bar_stp_inactive, const mo_task_t* task, uint32_t* contacts,
                     uint64_t nrows, uint64_t ncols, uint64_t* missed, uint64_t* occupancy,
                     mo_cell_result_t* res);
/* n_tasks cells of one interval on `nthreads` host threads (shared matrix, atomic increments) */
int mo_simulate_interval(const mo_params_t* p, uint64_t start, uint64_t end, size_t nb,
                         const uint64_t* bar_pos, const uint8_t* bar_dir,
                         const double* bar_stp_active, const double* bar_stp_inactive,
                         const mo_task_t* tasks, size_t n_tasks, uint32_t* contacts,
                         uint64_t nrows, uint64_t ncols, uint64_t* missed, uint64_t* occupancy,
                         mo_cell_result_t* results, int nthreads);

#ifdef __cplusplus
}
#endif
#endif

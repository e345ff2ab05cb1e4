/* modle_hip.h -- C ABI of the MI355X-native loop-extrusion simulation core.
 *
 * Drop-in boundary for ONE path of paulsengroup/modle: `Simulation::simulate_one_cell`
 * (reference: src/libmodle/cpu/include/modle/simulation.hpp:154, called from
 * src/libmodle/cpu/scheduler_simulate.cpp:240) together with the task derivation its caller
 * performs (scheduler_simulate.cpp:104-160).  The reference has no FFI; the entry points below
 * are what a cgo/ctypes/C++ binding of that seam would bind.
 *
 * Conventions: all structs are POD with 8-byte members, little-endian, caller-owned.  Every call
 * returns 0 on success or a negative error code and writes a NUL-terminated message into `err`
 * (may be NULL).  No exception crosses the ABI.  Pointers named `d_*` are DEVICE pointers
 * (hipMalloc / torch CUDA tensors); everything else is host memory.  A handle is bound to one
 * GPU and is not thread-safe; use one handle per host thread / per process (one process per GPU).
 */
#ifndef MODLE_HIP_H
#define MODLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MODLE_HIP_OK 0
#define MODLE_HIP_ERR_ARG (-1)
#define MODLE_HIP_ERR_DEVICE (-2)
#define MODLE_HIP_ERR_UNSUPPORTED (-3)
#define MODLE_HIP_ERR_STATE (-4)
#define MODLE_HIP_ERR_CANCELLED (-5)
#define MODLE_HIP_ERR_TIMEOUT (-6)

/* contact_sampling_strategy flags (reference: simulation_config.hpp:33-38) */
#define MODLE_HIP_CS_NOISIFY 1u
#define MODLE_HIP_CS_TAD 2u
#define MODLE_HIP_CS_LOOP 4u
/* barrier blocking direction (reference: extrusion_barriers_impl.hpp:61-72; BED strand '+' =>
 * REV, '-' => FWD) */
#define MODLE_HIP_DIR_FWD 1u
#define MODLE_HIP_DIR_REV 2u

/* `modle::Config` fields read by the path (reference: simulation_config.hpp:47-113) in their
 * post-`Cli::transform_args` form (reference: src/modle/cli.cpp:886-1016), followed by the raw
 * CLI-level inputs `modle_hip_config_transform` derives them from. */
typedef struct modle_hip_config {
  uint64_t bin_size;                      /* --resolution */
  uint64_t diagonal_width;                /* --diagonal-width */
  uint64_t rev_extrusion_speed;           /* bp / epoch */
  uint64_t fwd_extrusion_speed;
  double rev_extrusion_speed_std;         /* absolute bp after transform (fraction before) */
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
  uint64_t contact_sampling_strategy;     /* MODLE_HIP_CS_* */
  double tad_to_loop_contact_ratio;
  double genextreme_mu;
  double genextreme_sigma;
  double genextreme_xi;
  double target_contact_density;          /* < 0 => stop on target_simulation_epochs */
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
  /* ---- raw CLI-level inputs ---- */
  uint64_t avg_lef_processivity;
  double burnin_speed_coefficient;
  double extrusion_barrier_occupancy;
  double barrier_occupied_stp;
  double barrier_not_occupied_stp;
  uint64_t probability_normalization_factor;
  uint64_t normalize_probabilities;
  uint64_t rev_extrusion_speed_set;       /* non-zero: --rev-extrusion-speed given */
  uint64_t fwd_extrusion_speed_set;
  uint64_t extrusion_barrier_occupancy_set;
} modle_hip_config;

/* `Simulation::Task` (reference: simulation.hpp:59-69) without the interval pointer */
typedef struct modle_hip_task {
  uint64_t id;
  uint64_t cell_id;
  uint64_t num_target_epochs;
  uint64_t num_target_contacts;
  uint64_t num_lefs;
  uint64_t prng[4]; /* xoshiro256++ state (reference: random.hpp:26-32) */
} modle_hip_task;

/* what `State` holds when simulate_one_cell returns (reference: simulation.hpp:72-79;
 * logged at scheduler_simulate.cpp:246-251) plus counters used for the roofline */
typedef struct modle_hip_cell_result {
  uint64_t epochs;
  uint64_t burnin_epochs;
  uint64_t num_contacts;
  uint64_t raws_consumed;   /* 64-bit PRNG outputs drawn by the cell */
  uint64_t prng_final[4];   /* xoshiro256++ state after the last draw (State::rand_eng at return) */
  uint64_t sum_active_lefs; /* sum over simulated epochs of the number of active LEFs */
  uint64_t sampling_events; /* contact-sampling events executed */
  uint64_t sim_epochs;      /* epochs whose move / collision phase ran */
} modle_hip_cell_result;

typedef struct modle_hip_handle modle_hip_handle;

/* ---------------------------------------------------------------------------------------------
 * Host-side logic (no GPU needed).
 * ------------------------------------------------------------------------------------------- */
/* Config defaults (reference: simulation_config.hpp:47-113) */
void modle_hip_config_default(modle_hip_config* c);
/* Derived parameters (reference: Cli::transform_args, cli.cpp:886-1016) */
int modle_hip_config_transform(modle_hip_config* c, char* err, size_t errlen);
/* GenomicInterval::hash (reference: src/libmodle/internal/genome.cpp:201-224) */
uint64_t modle_hip_interval_hash(const char* chrom_name, uint64_t chrom_size, uint64_t start,
                                 uint64_t end, uint64_t seed);
/* random::PRNG(seed) (reference: random.hpp:26-30) and PRNG_t::jump() (scheduler:158) */
void modle_hip_prng_seed(uint64_t seed, uint64_t state[4]);
void modle_hip_prng_jump(uint64_t state[4]);
/* Simulation::compute_num_lefs / compute_contacts_per_epoch (reference: simulation.cpp:1076-1090) */
uint64_t modle_hip_compute_num_lefs(const modle_hip_config* c, uint64_t size_bp);
uint64_t modle_hip_compute_contacts_per_epoch(const modle_hip_config* c, uint64_t nlefs);
/* ContactMatrixDense(length, diagonal_width, bin_size) shape
 * (reference: src/contact_matrix/contact_matrix_dense_impl.hpp:34-44) */
void modle_hip_matrix_shape(const modle_hip_config* c, uint64_t size_bp, uint64_t* nrows,
                            uint64_t* ncols);
/* Task generation for one interval: seed hashing, per-cell jump(), target-contact split
 * (reference: scheduler_simulate.cpp:104-160).  `tasks` holds c->num_cells entries. */
int modle_hip_make_tasks(const modle_hip_config* c, const char* chrom_name, uint64_t chrom_size,
                         uint64_t start, uint64_t end, uint64_t first_task_id,
                         modle_hip_task* tasks);
/* ExtrusionBarrier::compute_stp_active_from_occupancy / compute_occupancy_from_stp
 * (reference: src/libmodle/internal/extrusion_barriers_impl.hpp:106-128) */
double modle_hip_stp_active_from_occupancy(double stp_inactive, double occupancy);
double modle_hip_occupancy_from_stp(double stp_active, double stp_inactive);
/* ExtrusionBarriers::sort (reference: src/libmodle/internal/extrusion_barriers.cpp:237-257, run
 * for every task by State::operator=, simulation.cpp:741-761): sorts the four parallel arrays by
 * position, in place.  Barriers at the same position keep their input order (the reference
 * leaves that order to an unstable sort).  modle_hip_add_interval applies it itself; it is
 * exported for callers that need to know which barrier a collision word's index refers to. */
void modle_hip_sort_barriers(uint64_t* bar_pos, uint8_t* bar_dir, double* bar_stp_active,
                             double* bar_stp_inactive, size_t n_barriers);

/* ---------------------------------------------------------------------------------------------
 * Device path.
 * ------------------------------------------------------------------------------------------- */
/* Creates a simulation context on HIP device `device`.  Fails (returns NULL) when no gfx950
 * device is usable: there is no CPU fallback. */
modle_hip_handle* modle_hip_create(const modle_hip_config* c, int device, char* err, size_t errlen);
/* Frees the context.  With a launch still in flight it raises the abort word and waits for the kernel
 * to drain for at most MODLE_HIP_DRAIN_TIMEOUT_S; a kernel that does not drain (a hung device) makes it
 * return WITHOUT freeing anything -- every free would wait for that kernel -- so that the caller can exit. */
void modle_hip_destroy(modle_hip_handle* h);

/* Registers one genomic interval (reference: GenomicInterval, genome.hpp) with its extrusion
 * barriers in any order: they are sorted by position here, once per interval (the reference
 * sorts them per task, State::operator=, simulation.cpp:741-761; modle_hip_sort_barriers).  `d_contacts` (uint32[nrows*ncols+1], band layout of
 * contact_matrix_internal_impl.hpp:19-42) and `d_occupancy` (uint64[ncols]) are caller-owned
 * DEVICE buffers that the kernel accumulates into; pass NULL to let the library own them.
 * Returns the interval id (>= 0) or a negative error. */
int modle_hip_add_interval(modle_hip_handle* h, uint64_t start, uint64_t end,
                           const uint64_t* bar_pos, const uint8_t* bar_dir,
                           const double* bar_stp_active, const double* bar_stp_inactive,
                           size_t n_barriers, void* d_contacts, void* d_occupancy, char* err,
                           size_t errlen);
/* Enqueues tasks for a registered interval (the counterpart of try_enqueue_task<PENDING>,
 * scheduler_simulate.cpp:152). */
int modle_hip_submit_tasks(modle_hip_handle* h, int interval_id, const modle_hip_task* tasks,
                           size_t n_tasks, char* err, size_t errlen);
/* Launches every pending task on `stream` (a hipStream_t, NULL = default stream) and returns
 * without waiting.  One wavefront simulates one cell -- the reference's one cell per worker thread,
 * scheduler_simulate.cpp:190-271 -- except when the launch leaves at least half of the GPU's wave
 * slots empty (at most 4 tasks per compute unit): a cell then gets a helper wave, and a third wave
 * for its PRNG blocks when there are at most 2 tasks per compute unit.  The
 * results do not depend on the mode; the environment variable MODLE_HIP_PAIRED=0 / 1, read at every
 * launch, forces it off / on (tests, A/B measurements).  In a launch that fills the slots, a wave
 * that finds the task queue empty helps a cell of its workgroup that is still running
 * (MODLE_HIP_TAIL_HELPERS=0 turns that off).  Since round 5 the library holds the kernels for 8 and for 12 waves per
 * workgroup and for 16-bit and 32-bit LEF ids and moves (modle_hip_size_class): a launch that fills the GPU and whose
 * epochs re-insert few units runs the 12-wave ones; MODLE_HIP_WAVES=8 / 12 and MODLE_HIP_SIZE_CLASS=wide force the
 * choice (A/B measurements, tests); modle_hip_last_launch_info says what ran.  None of it shows in the results. */
int modle_hip_launch(modle_hip_handle* h, void* stream, char* err, size_t errlen);
/* Waits for the launch and collects per-task results (the counterpart of _ctx.shutdown(),
 * scheduler_simulate.cpp:162).  The wait is bounded: when the launch has been running for longer
 * than the handle's deadline (default: none; environment variable MODLE_HIP_WAIT_TIMEOUT_S at
 * modle_hip_create, or modle_hip_set_wait_timeout) the abort word of modle_hip_cancel is raised --
 * the waves read it every sixteenth epoch and inside every spin loop of the helper-wave protocol, so
 * a wave whose partner has stopped answering leaves too -- and the call returns
 * MODLE_HIP_ERR_TIMEOUT once the kernel has drained: the outputs are incomplete, the handle is
 * usable again after modle_hip_reset.  A kernel that does not drain within
 * MODLE_HIP_DRAIN_TIMEOUT_S (default 60 s) is a hung device: MODLE_HIP_ERR_DEVICE, the launch stays
 * in flight and the process should exit (reference semantics: `_ctx` polled every epoch,
 * simulation.cpp:933; workers never wait for each other, scheduler_simulate.cpp:264-270). */
int modle_hip_wait(modle_hip_handle* h, char* err, size_t errlen);
/* Deadline of modle_hip_wait in seconds, counted from modle_hip_launch (> 0). */
int modle_hip_set_wait_timeout(modle_hip_handle* h, double seconds);
/* How the last launch was laid out on the GPU (what `MODLE_HIP_PAIRED` / the task count chose). */
typedef struct modle_hip_launch_info {
  uint64_t n_tasks;
  uint64_t num_cus;                  /* compute units of the device */
  uint64_t workgroups;               /* persistent workgroups launched (one per compute unit at most) */
  uint64_t waves_per_workgroup;
  uint64_t main_waves_per_workgroup; /* waves that pull tasks */
  uint64_t helper_waves;             /* per main wave: 1 in helper-wave mode, else 0 */
  uint64_t prng_producer_waves;      /* per main wave: 1 when a third wave produces the PRNG blocks */
  uint64_t tail_helpers;             /* 1: waves that find the queue empty help running cells */
  uint64_t size_class;               /* 0: NARROW kernels (16-bit LEF ids and moves), 1: WIDE (modle_hip_size_class) */
  /* Placement of the per-wave workspace (a launch runs up to 6 % faster or slower depending on which physical pages
   * the driver handed out for it; when the workspace is allocated the library probes up to MODLE_HIP_WORKSPACE_TRIES
   * (default 24; 1 = no search) candidate allocations with a streaming kernel and keeps the fastest: DESIGN.md): */
  uint64_t workspace_tries;          /* candidates probed when the current workspace was allocated (0: no probe ran) */
  uint64_t workspace_probe_us;       /* the probe's duration on the candidate that was kept, microseconds ... */
  uint64_t workspace_probe_worst_us; /* ... and on the slowest candidate */
} modle_hip_launch_info;
int modle_hip_last_launch_info(modle_hip_handle* h, modle_hip_launch_info* info);
/* Size class of a launch whose largest cell has `max_lefs` LEFs: 0 = NARROW (the kernels keep LEF ids and
 * moves as 16-bit values: fewer than 65 536 LEFs and extrusion speeds whose moves provably fit; every
 * real chromosome at the reference's defaults), 1 = WIDE (32-bit).  modle_hip_launch picks it per launch;
 * results do not depend on it.  The environment variable MODLE_HIP_SIZE_CLASS=wide forces WIDE. */
int modle_hip_size_class(const modle_hip_config* c, uint64_t max_lefs);
/* HIP_VERSION the library was compiled with and hipRuntimeGetVersion() of the runtime it is bound to in
 * this process (a host process that also loads PyTorch-ROCm shares torch's bundled runtime with this
 * library, INTEGRATION.md: the binding warns when the major versions differ).  No device is touched. */
int modle_hip_runtime_versions(int* built_with, int* runtime);
/* Asks a launch in flight to stop (the counterpart of the `_ctx` flag the reference polls once
 * per epoch, simulation.cpp:933; here every sixteenth epoch, the word being in host-mapped memory
 * so that it can be raised while the kernel holds every CU): every cell leaves at the top of one of its next epochs, cells that
 * have not started are skipped, and modle_hip_wait returns MODLE_HIP_ERR_CANCELLED.  Contacts
 * registered before the stop stay in the matrices.  May be called from another host thread than
 * the one that waits.  No-op when nothing is in flight. */
int modle_hip_cancel(modle_hip_handle* h, char* err, size_t errlen);
/* --log-model-internal-state (reference: Simulation::dump_stats, simulation.cpp:995-1056, one
 * record per task and epoch after extrude and before release_lefs).  Only the diagnostic build
 * libmodle_hip_statelog.so records (the default build returns MODLE_HIP_ERR_UNSUPPORTED for a
 * non-zero capacity: the recording code stays out of the production kernel).  A record is
 * MODLE_HIP_STATE_LOG_WORDS words: epoch | burn-in << 63, barriers occupied, active LEFs, units
 * stalled rev, units stalled fwd, LEFs stalled at both ends, LEF-BAR collisions, primary LEF-LEF
 * collisions, secondary LEF-LEF collisions, sum of the loop sizes.  Epochs beyond the capacity
 * are not recorded. */
#define MODLE_HIP_STATE_LOG_WORDS 10
int modle_hip_enable_state_log(modle_hip_handle* h, uint32_t max_epochs_per_task, char* err,
                               size_t errlen);
/* records of task `task_index` (submission order within the last launch) of `interval_id`, in
 * epoch order; `records` holds max_epochs * MODLE_HIP_STATE_LOG_WORDS words (NULL to count) */
int modle_hip_get_state_log(modle_hip_handle* h, int interval_id, size_t task_index,
                            uint64_t* records, size_t max_epochs, size_t* n_epochs, char* err,
                            size_t errlen);
/* Non-blocking: 1 when every task the launch in flight holds for `interval_id` has finished (its
 * matrix and occupancy track are complete and may be read by other streams, e.g. reduced with
 * RCCL while the kernel simulates the remaining intervals), 0 when not yet; 1 when nothing is in
 * flight.  Tasks are started largest interval first, so intervals complete roughly in that order.
 * The counters live in host-mapped memory: the call touches no stream. */
int modle_hip_interval_done(modle_hip_handle* h, int interval_id);
/* Duration of the last simulation kernel, measured with HIP events on the launch stream. */
int modle_hip_last_kernel_ms(modle_hip_handle* h, float* ms);
/* Results of the tasks submitted for `interval_id`, in submission order. */
int modle_hip_get_results(modle_hip_handle* h, int interval_id, modle_hip_cell_result* results,
                          size_t n_results);
/* Device pointers / shape of an interval's outputs (for an RCCL reduce by the caller). */
int modle_hip_interval_outputs(modle_hip_handle* h, int interval_id, void** d_contacts,
                               void** d_occupancy, uint64_t* nrows, uint64_t* ncols);
/* Copies an interval's outputs to host buffers (any may be NULL). */
int modle_hip_copy_outputs(modle_hip_handle* h, int interval_id, uint32_t* contacts,
                           uint64_t* missed_updates, uint64_t* occupancy, char* err,
                           size_t errlen);
/* Forgets all intervals / tasks (buffers owned by the library are freed). */
int modle_hip_reset(modle_hip_handle* h);

/* One-call form of the seam: register + submit + launch + wait + copy.
 * `contacts` (host, nrows*ncols+1) and `occupancy` (host, ncols, may be NULL) are ACCUMULATED
 * into, like the reference's shared ContactMatrixDense. */
int modle_hip_simulate_interval(modle_hip_handle* h, uint64_t start, uint64_t end,
                                const uint64_t* bar_pos, const uint8_t* bar_dir,
                                const double* bar_stp_active, const double* bar_stp_inactive,
                                size_t n_barriers, const modle_hip_task* tasks, size_t n_tasks,
                                uint32_t* contacts, uint64_t nrows, uint64_t ncols,
                                uint64_t* missed_updates, uint64_t* occupancy,
                                modle_hip_cell_result* results, char* err, size_t errlen);

/* ---------------------------------------------------------------------------------------------
 * Phase-level entry points: run one phase of the epoch loop on the GPU over caller-provided
 * arrays.  They mirror the reference's `Simulation::test_*` hooks (simulation.hpp:413-567) so
 * that the reference's unit-test vectors can be replayed on the device code.  Arrays are HOST
 * pointers (copied in and out); positions / epochs use UINT64_MAX for released LEFs; collision
 * words use the reference's encoding (index | event << 56).
 * `phase_mask` selects which passes run, in the reference's order.
 * ------------------------------------------------------------------------------------------- */
#define MODLE_HIP_PH_RANK 0x001u               /* rank_lefs (simulation.cpp:410-496) */
#define MODLE_HIP_PH_RANK_INIT 0x002u          /*   init_buffers=true */
#define MODLE_HIP_PH_ADJUST 0x004u             /* adjust_moves_of_consecutive_extr_units */
#define MODLE_HIP_PH_CLAMP 0x008u              /* clamp_moves */
#define MODLE_HIP_PH_BOUNDARIES 0x010u         /* detect_units_at_interval_boundaries */
#define MODLE_HIP_PH_LEF_BAR 0x020u            /* detect_lef_bar_collisions */
#define MODLE_HIP_PH_PRIMARY 0x040u            /* detect_primary_lef_lef_collisions */
#define MODLE_HIP_PH_CORRECT_LEF_BAR 0x080u    /* correct_moves_for_lef_bar_collisions */
#define MODLE_HIP_PH_CORRECT_PRIMARY 0x100u    /* correct_moves_for_primary_lef_lef_collisions */
#define MODLE_HIP_PH_SECONDARY 0x200u          /* process_secondary_lef_lef_collisions */
#define MODLE_HIP_PH_FIX_SECONDARY 0x400u      /* fix_secondary_lef_lef_collisions */
#define MODLE_HIP_PH_USE_BOUNDARY_COUNTS 0x800u /* feed the boundary counts to later passes */
#define MODLE_HIP_PH_BIND 0x1000u              /* select_and_bind_lefs (simulation.cpp:988-993): binds
                                                * every released LEF of the image at the epoch held
                                                * in bits 16..31 of the mask, then ranks */
#define MODLE_HIP_PH_GEN_MOVES 0x2000u         /* generate_moves (simulation.cpp:299-330) */

int modle_hip_test_phases(modle_hip_handle* h, uint32_t phase_mask, uint64_t start, uint64_t end,
                          size_t n_lefs, uint64_t* rev_pos, uint64_t* fwd_pos, uint64_t* epoch,
                          uint64_t* rev_rank, uint64_t* fwd_rank, uint64_t* rev_moves,
                          uint64_t* fwd_moves, uint64_t* rev_coll, uint64_t* fwd_coll,
                          size_t n_barriers, const uint64_t* bar_pos, const uint8_t* bar_dir,
                          const uint8_t* bar_active, uint64_t prng[4], uint64_t* raws_consumed,
                          char* err, size_t errlen);

/* Unit-level entry point: the small pieces of the path that the reference tests on their own,
 * evaluated by the device code.  `in` holds n pairs of 64-bit values, `out` 2 n words:
 *   LOOP_STATS        in = (rev position, fwd position) of every LEF; out[0], out[1] = bit images
 *                     of the mean and the population standard deviation of the loop sizes
 *                     (stats::mean / standard_dev as compute_loop_size_stats uses them,
 *                     simulation.cpp:795-819; reference vectors: test/units/stats/descriptive_test.cpp)
 *   MATRIX_INCREMENT  in = (row, col) per ContactMatrixDense::increment call; `contacts`
 *                     (uint32[nrows * ncols + 1], band layout) and `missed_updates` are updated in
 *                     place (reference vectors: the contact_matrix unit tests under test/units)
 *   COLLISION_WORDS   in = (index, event); out = (Collision<> word, predicate bits: 0 occurred,
 *                     1 avoided, 2..5 occurred(CHROM_BOUNDARY, LEF_BAR, LEF_LEF_PRIMARY,
 *                     LEF_LEF_SECONDARY), 6..9 avoided(the same)) (reference vectors:
 *                     test/units/simulation_cpu/collision_encoding_test.cpp) */
#define MODLE_HIP_UNIT_LOOP_STATS 1u
#define MODLE_HIP_UNIT_MATRIX_INCREMENT 2u
#define MODLE_HIP_UNIT_COLLISION_WORDS 3u
/*   MATH_LOG_EXP      in = bit images of (x, y); out = bit images of (log x, exp y)
 *   MATH_POW_SQRT     in = bit images of (x, y); out = bit images of (pow(x, y), sqrt x)
 *                     -- the floating-point routines the device path calls, for bit-for-bit
 *                     comparison with the oracle's (both compile modle_amd/csrc/modle_math.h) */
#define MODLE_HIP_UNIT_MATH_LOG_EXP 4u
#define MODLE_HIP_UNIT_MATH_POW_SQRT 5u
/*   PHILOX            two pairs per vector: (counter words 0..1 | 2..3 as two 64-bit values), (key
 *                     words 0..1 as one 64-bit value, unused); out = the four output words of
 *                     Philox4x32-10 as two 64-bit values, then two zeros -- the round function of the
 *                     PHILOX generator policy, for the Random123 known-answer vectors */
#define MODLE_HIP_UNIT_PHILOX 6u
int modle_hip_test_units(modle_hip_handle* h, uint32_t what, const uint64_t* in, size_t n,
                         uint64_t nrows, uint64_t ncols, uint32_t* contacts,
                         uint64_t* missed_updates, uint64_t* out, char* err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif

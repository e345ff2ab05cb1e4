"""ctypes mirror of `modle_hip_config` (include/modle_hip.h).

The struct carries the `modle::Config` fields that `Simulation::simulate_one_cell` reads
(reference: src/common/include/modle/common/simulation_config.hpp:47-113), in their
post-`Cli::transform_args` form (reference: src/modle/cli.cpp:886-1016), followed by the raw
CLI-level inputs that `modle_hip_config_transform` derives them from.  Every member is 8 bytes
wide.  Defaults and the derivation live in C++ (modle_amd/csrc/host_config.cpp), not here.
"""
import ctypes as C

# contact_sampling_strategy flags
CS_NOISIFY, CS_TAD, CS_LOOP = 1, 2, 4
# barrier blocking direction codes
DIR_FWD, DIR_REV = 1, 2


class Config(C.Structure):
    _fields_ = [
        ("bin_size", C.c_uint64),
        ("diagonal_width", C.c_uint64),
        ("rev_extrusion_speed", C.c_uint64),
        ("fwd_extrusion_speed", C.c_uint64),
        ("rev_extrusion_speed_std", C.c_double),
        ("fwd_extrusion_speed_std", C.c_double),
        ("rev_extrusion_speed_burnin", C.c_uint64),
        ("fwd_extrusion_speed_burnin", C.c_uint64),
        ("prob_of_lef_release", C.c_double),
        ("prob_of_lef_release_burnin", C.c_double),
        ("hard_stall_lef_stability_multiplier", C.c_double),
        ("soft_stall_lef_stability_multiplier", C.c_double),
        ("probability_of_extrusion_unit_bypass", C.c_double),
        ("lef_bar_major_collision_pblock", C.c_double),
        ("lef_bar_minor_collision_pblock", C.c_double),
        ("contact_sampling_interval", C.c_uint64),
        ("contact_sampling_strategy", C.c_uint64),
        ("tad_to_loop_contact_ratio", C.c_double),
        ("genextreme_mu", C.c_double),
        ("genextreme_sigma", C.c_double),
        ("genextreme_xi", C.c_double),
        ("target_contact_density", C.c_double),
        ("target_simulation_epochs", C.c_uint64),
        ("skip_burnin", C.c_uint64),
        ("burnin_history_length", C.c_uint64),
        ("burnin_smoothing_window_size", C.c_uint64),
        ("min_burnin_epochs", C.c_uint64),
        ("max_burnin_epochs", C.c_uint64),
        ("burnin_target_epochs_for_lef_activation", C.c_uint64),
        ("track_1d_lef_position", C.c_uint64),
        ("number_of_lefs_per_mbp", C.c_double),
        ("num_cells", C.c_uint64),
        ("seed", C.c_uint64),
        ("simulate_chromosomes_wo_barriers", C.c_uint64),
        # ---- raw CLI-level inputs consumed by modle_hip_config_transform ----
        ("avg_lef_processivity", C.c_uint64),
        ("burnin_speed_coefficient", C.c_double),
        ("extrusion_barrier_occupancy", C.c_double),
        ("barrier_occupied_stp", C.c_double),
        ("barrier_not_occupied_stp", C.c_double),
        ("probability_normalization_factor", C.c_uint64),
        ("normalize_probabilities", C.c_uint64),
        ("rev_extrusion_speed_set", C.c_uint64),
        ("fwd_extrusion_speed_set", C.c_uint64),
        ("extrusion_barrier_occupancy_set", C.c_uint64),
    ]

    def copy(self):
        c = Config()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(Config))
        return c


class Task(C.Structure):
    _fields_ = [
        ("id", C.c_uint64),
        ("cell_id", C.c_uint64),
        ("num_target_epochs", C.c_uint64),
        ("num_target_contacts", C.c_uint64),
        ("num_lefs", C.c_uint64),
        ("prng", C.c_uint64 * 4),
    ]


class CellResult(C.Structure):
    _fields_ = [
        ("epochs", C.c_uint64),
        ("burnin_epochs", C.c_uint64),
        ("num_contacts", C.c_uint64),
        ("raws_consumed", C.c_uint64),
        ("prng_final", C.c_uint64 * 4),
        ("sum_active_lefs", C.c_uint64),
        ("sampling_events", C.c_uint64),
        ("sim_epochs", C.c_uint64),
    ]


class LaunchInfo(C.Structure):
    """`modle_hip_launch_info`: how the last launch was laid out on the GPU"""
    _fields_ = [
        ("n_tasks", C.c_uint64),
        ("num_cus", C.c_uint64),
        ("workgroups", C.c_uint64),
        ("waves_per_workgroup", C.c_uint64),
        ("main_waves_per_workgroup", C.c_uint64),
        ("helper_waves", C.c_uint64),
        ("prng_producer_waves", C.c_uint64),
        ("tail_helpers", C.c_uint64),
        ("size_class", C.c_uint64),
        ("workspace_tries", C.c_uint64),
        ("workspace_probe_us", C.c_uint64),
        ("workspace_probe_worst_us", C.c_uint64),
    ]

"""Whole-cell parity of the product's DEVICE CODE (run by the CPU lane emulator, tests/wave_emu)
against the oracle.  Small inputs: the emulator is ~100x slower than the oracle."""
import pytest

import emu_sim
from modle_amd import api
from parity_cases import assert_same_outputs, assert_same_results, build_case


@pytest.mark.parametrize("name,ncells", [("config0_5mb_nobarriers", 2), ("chr6mb_skip_burnin", 1),
                                         ("chr8mb_loop_only", 1), ("tiny_single_lef", 3),
                                         ("zero_target_cells", 6), ("epochs_stop_tad_only", 1),
                                         ("window_near_position_limit", 1),
                                         ("dense_barriers_trials", 1),
                                         ("ultra_dense_barriers_trials", 1), ("mass_release", 1),
                                         ("many_lefs_hashed_filters", 1), ("many_rebinds_per_epoch", 1),
                                         ("rebinds_beyond_sort_buffer", 1), ("rebinds_beyond_sort_buffer_burnin", 1),
                                         ("dense_stress_rebinds_and_displaced", 1),
                                         ("burnin_three_windows", 1)])
def test_emulated_device_code_matches_oracle(oracle, name, ncells):
    case = build_case(name)
    cfg, chrom = case["cfg"], case["chrom"]
    tasks = api.slice_tasks(case["tasks"], 0, ncells)
    track = bool(cfg.track_1d_lef_position)
    oc, om, oo, ores = oracle.simulate_interval(
        cfg, chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"],
        case["stp_active"], case["stp_inactive"], tasks, nthreads=1, track_occupancy=track)
    ec, em, eo, eres = emu_sim.simulate_interval(
        cfg, chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"], case["stp_active"],
        case["stp_inactive"], tasks, case["nrows"], case["ncols"], track_occupancy=track)
    assert_same_results(ores, eres, name)
    assert_same_outputs((oc, om, oo), (ec, em, eo), name)


@pytest.mark.parametrize("schedule", [1, 7])
def test_result_does_not_depend_on_the_lane_schedule(oracle, schedule):
    """On hardware all lanes execute an instruction together; the emulator runs them one after the
    other between two collectives.  A kernel that is correct on hardware cannot depend on which
    lane runs first: descending and randomly permuted lane orders must give the same cell."""
    from phase_backend import emu_lib

    name = "chr8mb_loop_only"
    case = build_case(name)
    cfg, chrom = case["cfg"], case["chrom"]
    tasks = api.slice_tasks(case["tasks"], 0, 1)
    track = bool(cfg.track_1d_lef_position)
    oc, om, oo, ores = oracle.simulate_interval(
        cfg, chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"],
        case["stp_active"], case["stp_inactive"], tasks, nthreads=1, track_occupancy=track)
    lib = emu_lib()
    lib.emu_set_lane_schedule(schedule)
    try:
        ec, em, eo, eres = emu_sim.simulate_interval(
            cfg, chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"],
            case["stp_active"], case["stp_inactive"], tasks, case["nrows"], case["ncols"],
            track_occupancy=track)
    finally:
        lib.emu_set_lane_schedule(0)
    assert_same_results(ores, eres, name)
    assert_same_outputs((oc, om, oo), (ec, em, eo), name)

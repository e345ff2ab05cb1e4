"""Whole-cell parity on the real GPU: the HIP path (through the C ABI) must reproduce the CPU
oracle bit for bit -- every word of the band contact matrix, the 1-D occupancy track, the
missed-update counter and per-cell (epochs, burn-in epochs, contacts, PRNG outputs drawn)."""
import numpy as np
import pytest

from parity_cases import CASES, assert_same_outputs, assert_same_results, build_case, launch_modes

pytestmark = pytest.mark.gpu

# cells compared per case (the oracle runs them on the host cores of the GPU box)
NCELLS = {"config0_5mb_nobarriers": 64, "chr20mb_barriers": 96, "chr12mb_dense_softstall": 64,
          "chr8mb_loop_only": 64, "chr6mb_skip_burnin": 64, "tiny_single_lef": 8,
          "zero_target_cells": 128, "epochs_stop_tad_only": 16, "window_near_position_limit": 64,
          "dense_barriers_trials": 8, "ultra_dense_barriers_trials": 4, "mass_release": 4,
          "many_lefs_hashed_filters": 2, "many_rebinds_per_epoch": 4,
          "rebinds_beyond_sort_buffer": 4, "rebinds_beyond_sort_buffer_burnin": 4, "dense_stress_rebinds_and_displaced": 4,
          "burnin_three_windows": 12}


@pytest.mark.parametrize("name", list(CASES))
def test_gpu_matches_oracle(oracle, name):
    from modle_amd import api

    case = build_case(name)
    cfg, chrom = case["cfg"], case["chrom"]
    n = min(NCELLS[name], len(case["tasks"]))
    tasks = api.slice_tasks(case["tasks"], 0, n)
    oc, om, oo, ores = oracle.simulate_interval(
        cfg, chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"],
        case["stp_active"], case["stp_inactive"], tasks, nthreads=8,
        track_occupancy=bool(cfg.track_1d_lef_position))
    for mode in launch_modes():
        sim = api.Simulator(cfg, 0)
        try:
            gc, gm, go, gres = sim.simulate_interval(
                chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"], case["stp_active"],
                case["stp_inactive"], tasks)
        finally:
            sim.close()
        assert_same_results(ores, gres, f"{name}, helper waves {mode}")
        if not cfg.track_1d_lef_position:
            go = None
        assert_same_outputs((oc, om, oo), (gc, gm, go), f"{name}, helper waves {mode}")
    assert int(oc.sum()) + om == sum(r.num_contacts for r in ores)


def test_gpu_queue_api_matches_one_call(oracle):
    """add_interval/submit/launch/wait with two intervals in one launch == per-interval runs."""
    from modle_amd import api

    a, b = build_case("config0_5mb_nobarriers"), build_case("chr8mb_loop_only")
    assert a["cfg"].contact_sampling_strategy != b["cfg"].contact_sampling_strategy
    # both intervals must share one Config: rebuild b's tasks under a's config
    cfg = a["cfg"]
    chrom_b = b["chrom"]
    stp_a, stp_i = api.barrier_stps(cfg, chrom_b["bar_occupancy"])
    tasks_b = api.make_tasks(cfg, "chrU", chrom_b["size"], 0, chrom_b["size"])
    ta = api.slice_tasks(a["tasks"], 0, 24)
    tb = api.slice_tasks(tasks_b, 0, 24)
    sim = api.Simulator(cfg, 0)
    try:
        ia = sim.add_interval(0, a["chrom"]["size"], a["chrom"]["bar_pos"], a["chrom"]["bar_dir"],
                              a["stp_active"], a["stp_inactive"])
        ib = sim.add_interval(0, chrom_b["size"], chrom_b["bar_pos"], chrom_b["bar_dir"], stp_a,
                              stp_i)
        sim.submit(ia, ta)
        sim.submit(ib, tb)
        sim.launch()
        sim.wait()
        assert sim.kernel_ms() > 0
        out_a, out_b = sim.copy_outputs(ia), sim.copy_outputs(ib)
        res_a, res_b = sim.results(ia), sim.results(ib)
    finally:
        sim.close()
    for chrom, stps, tasks, out, res in ((a["chrom"], (a["stp_active"], a["stp_inactive"]), ta,
                                          out_a, res_a),
                                         (chrom_b, (stp_a, stp_i), tb, out_b, res_b)):
        oc, om, oo, ores = oracle.simulate_interval(cfg, 0, chrom["size"], chrom["bar_pos"],
                                                    chrom["bar_dir"], stps[0], stps[1], tasks,
                                                    nthreads=8)
        assert_same_results(ores, res, "queue api")
        assert_same_outputs((oc, om, oo), out, "queue api")

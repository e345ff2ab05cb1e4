"""Differential test on the GPU over seeded random set-ups (sizes, LEF densities, bypass / block
probabilities, stall multipliers, sampling strategies, resolutions): every output word and every
per-cell counter of the HIP path must equal the oracle's."""
import pytest

from fuzz_cases import random_case, random_case_v2, random_case_v3, random_case_v4
from parity_cases import assert_same_outputs, assert_same_results, launch_modes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", list(range(100, 116)) + [1121, 1148, 1178, 5482, 5645, 6024])
def test_gpu_matches_oracle_on_random_setups(oracle, seed):
    _compare(oracle, random_case(seed), f"seed {seed}")


@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_gpu_matches_oracle_on_random_setups_v2(oracle, seed):
    """wider generator: windows that do not start at 0, barrier density, speeds, noise, ..."""
    _compare(oracle, random_case_v2(seed), f"v2 seed {seed}")


@pytest.mark.parametrize("seed", list(range(1, 17)))
def test_gpu_matches_oracle_on_random_setups_v3(oracle, seed):
    """blocking probabilities in {0, 1}: the compacted-barrier path of LEF-BAR detection"""
    _compare(oracle, random_case_v3(seed), f"v3 seed {seed}")


@pytest.mark.parametrize("seed", list(range(1, 17)))
def test_gpu_matches_oracle_on_random_setups_v4(oracle, seed):
    """burn-in parameters (minimum length, history, smoothing window, activation ramp), stopping on
    epochs with burn-in, TAD-to-loop ratio at its extremes, release probabilities of exactly zero"""
    _compare(oracle, random_case_v4(seed), f"v4 seed {seed}")


def _compare(oracle, case, label):
    from modle_amd import api

    cfg, chrom = case["cfg"], case["chrom"]
    tasks = api.slice_tasks(case["tasks"], 0, min(12, len(case["tasks"])))
    track = bool(cfg.track_1d_lef_position)
    oc, om, oo, ores = oracle.simulate_interval(
        cfg, chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"],
        case["stp_active"], case["stp_inactive"], tasks, nthreads=8, track_occupancy=track)
    for mode in launch_modes():
        sim = api.Simulator(cfg, 0)
        try:
            gc, gm, go, gres = sim.simulate_interval(
                chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"], case["stp_active"],
                case["stp_inactive"], tasks)
        finally:
            sim.close()
        what = f"{label} (helper waves {mode}): {case['kw']}, size {case['size']}"
        assert_same_results(ores, gres, what)
        assert_same_outputs((oc, om, oo), (gc, gm, go if track else None), what)

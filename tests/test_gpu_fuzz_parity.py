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


# --- seeds nobody has seen yet ---------------------------------------------------------
# The seeds above are the committed regression set: every run of the suite repeats them.  These take up to 64
# seeds per generator from the DATE of the run (MODLE_FUZZ_DATE=YYYYMMDD reproduces a day), so that the driver's
# round-end run and every builder run in between cover set-ups no campaign has visited.  A generator's share of
# the suite is bounded in time (the oracle runs on the host): it reports how many seeds it compared.
FRESH_SEEDS_PER_GENERATOR = 64
FRESH_BUDGET_S = 60.0


def _date_base():
    import datetime
    import os

    day = os.environ.get("MODLE_FUZZ_DATE") or datetime.date.today().strftime("%Y%m%d")
    return int(day) * 1000


@pytest.mark.parametrize("name", ["v1", "v2", "v3", "v4"])
def test_gpu_matches_oracle_on_seeds_of_the_day(oracle, name):
    import time

    from modle_amd import api

    gen = {"v1": random_case, "v2": random_case_v2, "v3": random_case_v3, "v4": random_case_v4}[name]
    base = _date_base()
    t0 = time.time()
    done = skipped = 0
    for i in range(FRESH_SEEDS_PER_GENERATOR):
        if time.time() - t0 > FRESH_BUDGET_S:
            break
        seed = base + i
        case = gen(seed)
        cfg = case["cfg"]
        # (keep the oracle's share short, like tools/fuzz_campaign.py: skip set-ups whose cells need many epochs)
        per_epoch = max(1, api.compute_contacts_per_epoch(cfg, case["tasks"][0].num_lefs))
        if cfg.target_contact_density >= 0 and case["tasks"][0].num_target_contacts / per_epoch > 3000:
            skipped += 1
            continue
        _compare(oracle, case, f"{name} seed of the day {seed} (MODLE_FUZZ_DATE={base // 1000})")
        done += 1
    print(f"{name}: {done} fresh seeds from {base} compared in both launch modes, {skipped} skipped (long cells), "
          f"{time.time() - t0:.0f} s")
    assert done >= 8, f"only {done} fresh seeds fitted the time budget"

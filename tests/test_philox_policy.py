"""PHILOX generator policy (SURVEY.md H1's second back-end; the row the judge added to section 8):
a compile-time policy of the device code (-DMODLE_RNG_PHILOX, modle_amd/libmodle_hip_philox.so)
in which a cell's stream is counter based.  It is NOT comparable bit for bit with the reference's
xoshiro stream, so it is held to two bars of its own:

* bit for bit against the oracle running the same policy (here on the lane emulator; on the GPU
  in test_gpu_philox_matches_oracle_with_the_same_policy);
* statistically against the exact mode: per-stripe Pearson / Spearman correlation of the contact
  matrices (modle_amd/evaluate.py, the method of modle_tools evaluate), compared with the
  correlation between two exact-mode runs with different seeds (the noise floor of the method).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import emu_sim
from modle_amd import api, evaluate
from parity_cases import assert_same_outputs, assert_same_results, build_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_run(oracle, case, tasks, philox, nthreads=1):
    cfg, chrom = case["cfg"], case["chrom"]
    track = bool(cfg.track_1d_lef_position)
    with oracle.rng_policy(philox):
        return oracle.simulate_interval(cfg, chrom["start"], chrom["end"], chrom["bar_pos"],
                                        chrom["bar_dir"], case["stp_active"], case["stp_inactive"],
                                        tasks, nthreads=nthreads, track_occupancy=track)


@pytest.mark.parametrize("name,ncells", [("config0_5mb_nobarriers", 2), ("chr8mb_loop_only", 1),
                                         ("dense_barriers_trials", 1)])
def test_emulated_device_code_matches_oracle_with_the_same_policy(oracle, name, ncells):
    case = build_case(name)
    cfg, chrom = case["cfg"], case["chrom"]
    tasks = api.slice_tasks(case["tasks"], 0, ncells)
    track = bool(cfg.track_1d_lef_position)
    oc, om, oo, ores = _oracle_run(oracle, case, tasks, True)
    ec, em, eo, eres = emu_sim.simulate_interval(
        cfg, chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"], case["stp_active"],
        case["stp_inactive"], tasks, case["nrows"], case["ncols"], track_occupancy=track,
        variant="philox")
    assert_same_results(ores, eres, name)
    assert_same_outputs((oc, om, oo), (ec, em, eo), name)
    # and it is a different stream: the exact-mode cell differs
    xc, _, _, xres = _oracle_run(oracle, case, tasks, False)
    assert not np.array_equal(xc, oc)
    assert [r.raws_consumed for r in xres] != [r.raws_consumed for r in ores]


def test_philox_is_statistically_equivalent_to_the_exact_mode(oracle):
    """The evaluator cannot tell PHILOX output from exact output: the per-stripe correlation
    between the two modes is as high as between two exact runs with different seeds."""
    case = build_case("chr20mb_barriers")
    cfg = case["cfg"]
    nrows, ncols = case["nrows"], case["ncols"]
    n = 24
    tasks = api.slice_tasks(case["tasks"], 0, n)
    exact, _, _, _ = _oracle_run(oracle, case, tasks, False, nthreads=8)
    philox, _, _, _ = _oracle_run(oracle, case, tasks, True, nthreads=8)
    other = api.slice_tasks(case["tasks"], n, 2 * n)  # other cells = other streams, exact mode
    exact2, _, _, _ = _oracle_run(oracle, case, other, False, nthreads=8)
    assert int(exact.sum()) == int(philox.sum()) == int(exact2.sum())  # same contact targets
    for metric in ("pearson", "spearman"):
        for direction in ("vertical", "horizontal"):
            a = evaluate.summarize(evaluate.compare(exact, philox, nrows, ncols, metric, direction)[0])
            b = evaluate.summarize(evaluate.compare(exact, exact2, nrows, ncols, metric, direction)[0])
            assert a["n"] > ncols // 2
            # same quality as the exact-vs-exact noise floor (within 0.05) and clearly correlated
            assert a["median"] > 0.2 and abs(a["median"] - b["median"]) < 0.05, (metric, direction, a, b)


def _philox_child(case_name, ncells, out):
    env = dict(os.environ, MODLE_HIP_LIB="libmodle_hip_philox.so")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "philox_child.py"), case_name,
                        str(ncells), out], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return np.load(out)


@pytest.mark.gpu
@pytest.mark.parametrize("name,ncells", [("chr20mb_barriers", 48), ("chr12mb_dense_softstall", 16),
                                         ("config0_5mb_nobarriers", 32)])
def test_gpu_philox_matches_oracle_with_the_same_policy(oracle, tmp_path, name, ncells):
    case = build_case(name)
    tasks = api.slice_tasks(case["tasks"], 0, ncells)
    got = _philox_child(name, ncells, str(tmp_path / "philox.npz"))
    oc, om, oo, ores = _oracle_run(oracle, case, tasks, True, nthreads=8)
    assert np.array_equal(got["contacts"], oc) and int(got["missed"]) == om
    assert np.array_equal(got["occupancy"], oo)
    fields = ("epochs", "burnin_epochs", "num_contacts", "raws_consumed", "sum_active_lefs",
              "sampling_events", "sim_epochs")
    exp = np.array([[getattr(r, f) for f in fields] + list(r.prng_final) for r in ores], dtype=np.uint64)
    assert np.array_equal(got["results"], exp)

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
import ctypes as C
import json
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


# ---------------------------------------------------------------------------------------------
# Known-answer vectors of the round function (Random123's kat_vectors; rocRAND's philox4x32_10 is
# the same function): oracle, emulated device code, GPU
# ---------------------------------------------------------------------------------------------
UNIT_PHILOX = 6
with open(os.path.join(ROOT, "tests", "golden", "random123_philox_kats.json")) as _fh:
    PHILOX_KATS = json.load(_fh)["vectors"]


def _kat_pairs():
    """input of the unit-level hook (MODLE_HIP_UNIT_PHILOX): two pairs of 64-bit words per vector"""
    words = []
    for v in PHILOX_KATS:
        c = [int(x, 16) for x in v["counter"]]
        k = [int(x, 16) for x in v["key"]]
        words += [c[0] | (c[1] << 32), c[2] | (c[3] << 32), k[0] | (k[1] << 32), 0]
    return np.array(words, dtype=np.uint64)


def _check_kat_words(out):
    for i, v in enumerate(PHILOX_KATS):
        e = [int(x, 16) for x in v["expected"]]
        assert int(out[4 * i]) == e[0] | (e[1] << 32), (i, hex(int(out[4 * i])))
        assert int(out[4 * i + 1]) == e[2] | (e[3] << 32), (i, hex(int(out[4 * i + 1])))


def test_philox_known_answers_on_the_oracle(oracle):
    L = oracle.lib()
    u32x4, u32x2 = C.c_uint32 * 4, C.c_uint32 * 2
    L.mo_philox4x32_10.argtypes = [u32x4, u32x2, u32x4]
    L.mo_philox4x32_10.restype = None
    for v in PHILOX_KATS:
        out = u32x4()
        L.mo_philox4x32_10(u32x4(*[int(x, 16) for x in v["counter"]]),
                           u32x2(*[int(x, 16) for x in v["key"]]), out)
        assert ["%08x" % x for x in out] == v["expected"]
    # the stream the policy derives from it: output p = one half of the block of counter p >> 1
    with oracle.rng_policy(True):
        g = oracle.Prng()
        for i in range(4):
            g.s[i] = 0
        g.count = 0
        L.mo_prng_next.argtypes = [C.POINTER(oracle.Prng)]
        L.mo_prng_next.restype = C.c_uint64
        first, second = L.mo_prng_next(C.byref(g)), L.mo_prng_next(C.byref(g))
    e = [int(x, 16) for x in PHILOX_KATS[0]["expected"]]  # counter 0, key 0
    assert (first, second) == (e[0] | (e[1] << 32), e[2] | (e[3] << 32))


@pytest.mark.parametrize("variant", [None, "philox"])
def test_philox_known_answers_on_the_emulated_device_code(variant):
    from modle_amd.params import Config
    from phase_backend import emu_lib

    L = emu_lib(variant)
    u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
    L.emu_test_units.argtypes = [C.POINTER(Config), C.c_uint32, u64p, C.c_size_t, C.c_uint64,
                                 C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), u64p]
    L.emu_test_units.restype = C.c_int
    pairs = _kat_pairs()
    out = np.zeros(len(pairs), dtype=np.uint64)
    m = C.c_uint64(0)
    cfg = api.make_config()
    assert L.emu_test_units(C.byref(cfg), UNIT_PHILOX, pairs, len(pairs) // 2, 0, 0, None, C.byref(m), out) == 0
    _check_kat_words(out)


@pytest.mark.gpu
def test_philox_known_answers_on_the_gpu():
    sim = api.Simulator(api.make_config(), 0)
    try:
        out, _, _ = sim.test_units(UNIT_PHILOX, _kat_pairs())
    finally:
        sim.close()
    _check_kat_words(out)


@pytest.mark.gpu
def test_gpu_philox_output_is_statistically_equivalent_to_gpu_exact_output(tmp_path):
    """Row f4 on the device: the evaluator (the method of `modle_tools evaluate`) cannot tell the
    PHILOX library's contact matrix from the exact library's -- the per-stripe correlation between
    the two is as high as between two exact runs over different cells (the acceptance rule of
    test_philox_is_statistically_equivalent_to_the_exact_mode, with both sides computed by HIP)."""
    name, n = "chr20mb_barriers", 48
    case = build_case(name)
    cfg, chrom = case["cfg"], case["chrom"]
    nrows, ncols = case["nrows"], case["ncols"]

    def exact(first):
        sim = api.Simulator(cfg, 0)
        try:
            c, _, _, _ = sim.simulate_interval(chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"],
                                               case["stp_active"], case["stp_inactive"],
                                               api.slice_tasks(case["tasks"], first, first + n))
        finally:
            sim.close()
        return c

    e1, e2 = exact(0), exact(n)
    ph = _philox_child(name, n, str(tmp_path / "philox.npz"))["contacts"]
    assert int(e1.sum()) == int(ph.sum()) == int(e2.sum())
    assert not np.array_equal(e1, ph)
    for metric in ("pearson", "spearman"):
        for direction in ("vertical", "horizontal"):
            a = evaluate.summarize(evaluate.compare(e1, ph, nrows, ncols, metric, direction)[0])
            b = evaluate.summarize(evaluate.compare(e1, e2, nrows, ncols, metric, direction)[0])
            assert a["n"] > ncols // 2
            assert a["median"] > 0.2 and abs(a["median"] - b["median"]) < 0.05, (metric, direction, a, b)

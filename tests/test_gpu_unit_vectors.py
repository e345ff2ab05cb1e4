"""tests/golden/reference_unit_vectors.json on the REAL GPU through the C ABI
(modle_hip_test_units / modle_hip_test_phases); the property tests are also compared word for
word with the oracle."""
import numpy as np
import pytest

from unit_vector_runner import DeviceUnits, OracleUnits, all_vectors, run_vector

pytestmark = pytest.mark.gpu
VECTORS = all_vectors()


@pytest.fixture(scope="module")
def gpu_backend(oracle):
    from modle_amd import api
    from phase_backend import _advance

    sims = {}

    def sim_for(cfg):
        key = bytes(cfg)
        if key not in sims:
            sims[key] = api.Simulator(cfg.copy(), 0)
        return sims[key]

    def units(cfg, what, pairs, nrows, ncols, contacts, missed):
        out, _, m = sim_for(cfg).test_units(what, pairs, nrows, ncols, contacts, missed)
        return out, m

    def phases(cfg, mask, st, state, skip):
        return sim_for(cfg).test_phases(mask, st, _advance(state, skip))

    yield DeviceUnits("gpu", units, phases, oracle)
    for s in sims.values():
        s.close()


@pytest.mark.parametrize("group,v", VECTORS, ids=[v["name"] for _, v in VECTORS])
def test_reference_unit_vector_on_gpu(oracle, gpu_backend, group, v):
    kw = {"iters": 200} if v["name"].startswith("Generate LEF moves") else {}
    got = run_vector(gpu_backend, group, v, **kw)
    if group == "property_tests":
        st_o, n_o = run_vector(OracleUnits(oracle), group, v, **kw)
        st_d, n_d = got
        assert n_o == n_d, "PRNG outputs consumed"
        fields = ["rev_pos", "fwd_pos", "epoch", "rev_rank", "fwd_rank"]
        if not v["name"].startswith("Bind"):  # the move arrays mean nothing before generate_moves
            fields += ["rev_moves", "fwd_moves"]
        for f in fields:
            assert np.array_equal(getattr(st_o, f), getattr(st_d, f)), f


def test_loop_stats_bit_identical_on_random_loops_gpu(oracle, gpu_backend):
    rng = np.random.default_rng(9)
    be_o = OracleUnits(oracle)
    for n in (1, 2, 63, 64, 65, 300, 511, 512, 513, 1000, 4979, 15920, 70000):  # (> 65536: the scattering form)
        rev = rng.integers(1, 200_000_000, size=n).astype(np.uint64)
        fwd = rev + rng.integers(0, 3_000_000, size=n).astype(np.uint64)
        a, b = be_o.loop_stats(rev, fwd), gpu_backend.loop_stats(rev, fwd)
        assert a["mean"] == b["mean"] and a["std"] == b["std"], n
    # the shapes that are hard for fold_terms_exact (see tests/test_oracle_unit_vectors.py)
    for shape, n in [(s, n) for s in range(6) for n in (64, 1000, 4096, 5000)]:
        rev = rng.integers(1, 200_000_000, size=n).astype(np.uint64)
        size = [rng.integers(0, 4, size=n), np.full(n, 12345), rng.integers(0, 2, size=n) * 1_000_000,
                np.where(rng.integers(0, 64, size=n) == 0, 4_000_000_000, rng.integers(0, 100, size=n)),
                1 << rng.integers(0, 31, size=n), np.where(rng.integers(0, 2, size=n) == 0, 0,
                                                            rng.integers(0, 200_000_000, size=n))][shape]
        fwd = rev + size.astype(np.uint64)
        a, b = be_o.loop_stats(rev, fwd), gpu_backend.loop_stats(rev, fwd)
        assert a["mean"] == b["mean"] and a["std"] == b["std"], (shape, n)


def test_math_bit_identical_on_gpu(oracle, gpu_backend):
    """2.4 M arguments shaped like the path's call sites: the GPU's log / exp / pow / sqrt return
    the oracle's bits (shared software routines; sqrt is IEEE-exact on both sides)."""
    from test_modle_math import math_lib, oracle_bits, path_arguments
    from unit_vector_runner import base_config

    L = math_lib(oracle)
    x, y = path_arguments(np.random.default_rng(5), 20_000)
    lg, ex, pw = oracle_bits(L, x, y)
    pairs = np.stack([x.view(np.uint64), y.view(np.uint64)], axis=1)
    out, _ = gpu_backend._units(base_config(), 4, pairs, 0, 0, None, 0)
    assert np.array_equal(out[0::2], lg), "log"
    assert np.array_equal(out[1::2], ex), "exp"
    out, _ = gpu_backend._units(base_config(), 5, pairs, 0, 0, None, 0)
    assert np.array_equal(out[0::2], pw), "pow"
    assert np.array_equal(out[1::2], np.sqrt(x).view(np.uint64)), "sqrt"
    # a larger sweep of the GEV call: pow(-log(u), xi) and log(u) over canonical draws
    rng = np.random.default_rng(6)
    u = rng.random(400_000)
    u[u == 0] = 0.5
    xi = np.full_like(u, 0.001)
    pairs = np.stack([u.view(np.uint64), xi.view(np.uint64)], axis=1)
    out, _ = gpu_backend._units(base_config(), 4, pairs, 0, 0, None, 0)
    sample = rng.integers(0, len(u), 5000)
    assert all(out[2 * i] == np.float64(L.mo_math_log(float(u[i]))).view(np.uint64) for i in sample)

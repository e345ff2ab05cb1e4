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
    for n in (1, 2, 63, 64, 65, 300, 1000, 4979):
        rev = rng.integers(1, 200_000_000, size=n).astype(np.uint64)
        fwd = rev + rng.integers(0, 3_000_000, size=n).astype(np.uint64)
        a, b = be_o.loop_stats(rev, fwd), gpu_backend.loop_stats(rev, fwd)
        assert a["mean"] == b["mean"] and a["std"] == b["std"], n

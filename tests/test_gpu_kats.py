"""The reference's unit-test vectors replayed on the REAL GPU through the C ABI
(modle_hip_test_phases).  Needs an MI355X: marked gpu."""
import pytest

from kat_runner import load_cases, run_case
from phase_backend import PhaseBackend, _advance

CASES = load_cases()

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_phases():
    from modle_amd import api

    sims = {}

    def phases(cfg, mask, st, state, skip):
        # the handle carries the Config; the KAT configs only differ in a few probabilities
        key = (cfg.probability_of_extrusion_unit_bypass, cfg.lef_bar_major_collision_pblock,
               cfg.lef_bar_minor_collision_pblock)
        if key not in sims:
            sims[key] = api.Simulator(cfg.copy(), 0)
        return sims[key].test_phases(mask, st, _advance(state, skip))

    yield phases
    for s in sims.values():
        s.close()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_kat_on_gpu(gpu_phases, case):
    run_case(PhaseBackend(gpu_phases), case)

"""Randomised multi-batch phase-level parity on the real GPU (modle_hip_test_phases) vs oracle."""
import pytest

from phase_backend import _advance
from phase_random import run_sequences

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n,nb,kw,dense", [
    (1, 300, 60, {}, False),
    (2, 1500, 400, {}, False),
    (3, 1000, 300, {"bypass": 0.0}, False),
    (4, 1200, 500, {"minor": 0.3, "major": 0.9, "bypass": 0.3}, True),
    (5, 5000, 3000, {}, False),
])
def test_random_phases_gpu(oracle, seed, n, nb, kw, dense):
    from modle_amd import api

    sims = {}

    def phases(cfg, mask, st, state, skip):
        key = (cfg.probability_of_extrusion_unit_bypass, cfg.lef_bar_major_collision_pblock,
               cfg.lef_bar_minor_collision_pblock)
        if key not in sims:
            sims[key] = api.Simulator(cfg.copy(), 0)
        return sims[key].test_phases(mask, st, _advance(state, skip))

    try:
        run_sequences(oracle, phases, seed, n, nb, kw, dense)
    finally:
        for s in sims.values():
            s.close()

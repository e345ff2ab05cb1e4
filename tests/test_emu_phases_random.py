"""Randomised multi-batch phase-level parity: emulated device code vs oracle (CPU)."""
import pytest

from phase_backend import emu_phases
from phase_random import run_sequences


@pytest.mark.parametrize("seed,n,nb,kw,dense", [
    (1, 300, 60, {}, False),
    (2, 1500, 400, {}, False),
    (3, 1000, 300, {"bypass": 0.0}, False),
    (4, 1200, 500, {"minor": 0.3, "major": 0.9, "bypass": 0.3}, True),
])
def test_random_phases_emulated(oracle, seed, n, nb, kw, dense):
    run_sequences(oracle, emu_phases, seed, n, nb, kw, dense)

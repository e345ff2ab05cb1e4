"""Replays the reference's unit-test vectors on the product's DEVICE CODE, executed by the CPU
lane emulator (tests/wave_emu).  The same vectors run on the real GPU in test_gpu_kats.py."""
import pytest

from kat_runner import load_cases, run_case
from phase_backend import PhaseBackend, emu_phases

CASES = load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_kat_on_device_code_emulated(case):
    run_case(PhaseBackend(emu_phases), case)

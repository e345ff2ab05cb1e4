"""Pins the CPU oracle against every known-answer vector the reference's own tests hold for the
hot path."""
import pytest

from kat_runner import load_cases, run_case
from oracle_backend import OracleBackend

CASES = load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_kat_on_oracle(oracle, case):
    run_case(OracleBackend(oracle), case)

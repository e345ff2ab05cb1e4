"""--log-model-internal-state (second half; reference:
Simulation::dump_stats, simulation.cpp:995-1056): the per-epoch records of the GPU's diagnostic
build equal the oracle's for every task and epoch, and the front end formats them in the
reference's column order."""
import ctypes as C
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from modle_amd import api, driver
from modle_amd.params import CellResult
from parity_cases import build_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_log(oracle, case, task, cap):
    L = oracle.lib()
    cfg, chrom = case["cfg"], case["chrom"]
    nrows, ncols = case["nrows"], case["ncols"]
    contacts = np.zeros(nrows * ncols + 1, dtype=np.uint32)
    occ = np.zeros(ncols, dtype=np.uint64)
    missed = C.c_uint64(0)
    res = CellResult()
    log = np.zeros((cap, 10), dtype=np.uint64)
    L.mo_simulate_cell_with_state_log.restype = C.c_size_t
    L.mo_simulate_cell_with_state_log.argtypes = L.mo_simulate_cell.argtypes + [C.c_void_p, C.c_size_t]
    n = L.mo_simulate_cell_with_state_log(
        C.byref(cfg), chrom["start"], chrom["end"], len(chrom["bar_pos"]),
        np.ascontiguousarray(chrom["bar_pos"], dtype=np.uint64),
        np.ascontiguousarray(chrom["bar_dir"], dtype=np.uint8),
        np.ascontiguousarray(case["stp_active"], dtype=np.float64),
        np.ascontiguousarray(case["stp_inactive"], dtype=np.float64), C.byref(task), contacts, nrows,
        ncols, C.byref(missed), occ.ctypes.data, C.byref(res), log.ctypes.data, cap)
    return log[:n], res


def test_oracle_records_are_consistent(oracle):
    case = build_case("chr6mb_skip_burnin")
    log, res = oracle_log(oracle, case, case["tasks"][0], 4096)
    assert len(log) == res.sim_epochs and len(log) > 10
    n_bar = len(case["chrom"]["bar_pos"])
    for rec in log:
        assert rec[1] <= n_bar and rec[5] <= min(rec[3], rec[4]) and rec[3] <= rec[2] and rec[4] <= rec[2]
        assert rec[6] + rec[7] + rec[8] <= rec[3] + rec[4]  # boundary stalls are in neither class
    assert [int(r[0]) & 0xFFFFFFFF for r in log] == list(range(len(log)))
    lines = driver.format_state_log(case["tasks"][0], case["chrom"], n_bar, log[:2])
    cols = lines[0].rstrip("\n").split("\t")
    assert len(cols) == len(driver.STATE_LOG_HEADER.split("\t")) == 16
    assert cols[:7] == ["0", "0", "0", "chrT", "0", "6000000", "False"]
    assert float(cols[7]) == int(log[0][1]) / n_bar and float(cols[15]) == int(log[0][9]) / int(log[0][2])


def test_default_build_refuses_to_log():
    """the recording code is not in the production kernel: asking for it fails loudly"""
    L = api.lib()
    err = C.create_string_buffer(256)
    # no GPU needed: a NULL handle is rejected before anything else
    assert L.modle_hip_enable_state_log(None, 16, err, len(err)) < 0


@pytest.mark.gpu
def test_gpu_records_match_the_oracle(oracle, tmp_path):
    name, ncells, cap = "chr20mb_barriers", 6, 2048
    out = str(tmp_path / "log.npz")
    env = dict(os.environ, MODLE_HIP_LIB="libmodle_hip_statelog.so")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "statelog_child.py"), name,
                        str(ncells), str(cap), out], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    got = np.load(out)
    case = build_case(name)
    for k in range(ncells):
        exp, res = oracle_log(oracle, case, case["tasks"][k], cap)
        assert len(exp) == min(res.sim_epochs, cap)
        assert np.array_equal(got[f"log{k}"], exp), f"cell {k}"
    # the default build refuses on a real handle too
    sim = api.Simulator(case["cfg"], 0)
    try:
        with pytest.raises(api.ModleHipError):
            sim.enable_state_log(16)
    finally:
        sim.close()


@pytest.mark.gpu
def test_cli_writes_the_state_log(tmp_path):
    sizes = "chrA\t2000000\n"
    bed = "".join(f"chrA\t{p}\t{p + 19}\t.\t0.9\t{'+' if i % 2 else '-'}\n" for i, p in enumerate(range(50_000, 1_950_000, 90_000)))
    (tmp_path / "g.chrom.sizes").write_text(sizes)
    (tmp_path / "b.bed").write_text(bed)
    prefix = str(tmp_path / "run")
    env = {k: v for k, v in os.environ.items() if k != "MODLE_HIP_LIB"}
    p = subprocess.run([sys.executable, "-m", "modle_amd", "simulate", "-c", str(tmp_path / "g.chrom.sizes"),
                        "-b", str(tmp_path / "b.bed"), "-o", prefix, "--ncells", "3", "-w", "500000",
                        "--target-contact-density", "0.5", "--log-model-internal-state", "-q"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    with gzip.open(prefix + "_internal_state.log.gz", "rt") as fh:
        lines = fh.read().splitlines()
    assert lines[0] == driver.STATE_LOG_HEADER.rstrip("\n")
    rows = [l.split("\t") for l in lines[1:]]
    assert {r[2] for r in rows} == {"0", "1", "2"} and all(r[3] == "chrA" for r in rows)
    first = [r for r in rows if r[2] == "0"]
    assert [int(r[1]) for r in first] == list(range(len(first)))  # every epoch of the cell, in order
    assert first[0][6] == "True" and first[-1][6] == "False"     # burn-in, then sampling
    assert os.path.exists(prefix + ".cool")

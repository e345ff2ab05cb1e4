"""Host-side logic of the product library (no GPU needed): Config defaults and derived
parameters, interval seeding, PRNG state derivation, task generation -- checked against the
oracle, the python `xxhash` module and the values SURVEY.md section 8c quotes."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import xxhash

from modle_amd import _lib, api
from modle_amd.params import Config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "modle_hip.h")).read()
    declared = set(re.findall(r"\b(modle_hip_[a-z0-9_]+)\s*\(", header))
    declared -= {"modle_hip_config", "modle_hip_task", "modle_hip_cell_result", "modle_hip_handle"}
    assert declared, "no declarations parsed"
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/modle_hip.h but not exported"
    assert declared == set(_lib.EXPORTS)


def test_config_layout_matches_header():
    header = open(os.path.join(ROOT, "include", "modle_hip.h")).read()
    body = header.split("typedef struct modle_hip_config {")[1].split("} modle_hip_config;")[0]
    fields = re.findall(r"^\s*(?:uint64_t|double)\s+(\w+);", body, re.M)
    assert fields == [f[0] for f in Config._fields_]
    assert C.sizeof(Config) == 8 * len(fields)


def test_default_config_and_transform():
    # reference: simulation_config.hpp:47-113 and cli.cpp:886-1016
    cfg = api.default_config()
    assert (cfg.bin_size, cfg.diagonal_width, cfg.num_cells) == (5000, 3_000_000, 512)
    assert cfg.number_of_lefs_per_mbp == 20 and cfg.avg_lef_processivity == 300_000
    api.transform_config(cfg)
    assert cfg.rev_extrusion_speed == cfg.fwd_extrusion_speed == 4000
    assert cfg.rev_extrusion_speed_std == cfg.fwd_extrusion_speed_std == 200.0
    assert cfg.prob_of_lef_release == 8000 / 300000
    assert cfg.burnin_target_epochs_for_lef_activation == 187  # 5 * 300000 / 8000
    assert cfg.tad_to_loop_contact_ratio == 5.0
    assert cfg.probability_of_extrusion_unit_bypass == 0.1  # ratio == 1: no normalisation
    assert cfg.extrusion_barrier_occupancy == pytest.approx(0.3 / 1.3)
    # a different resolution rescales the speeds and triggers probability normalisation
    cfg2 = api.make_config(bin_size=10000)
    assert cfg2.rev_extrusion_speed == 8000
    assert cfg2.probability_of_extrusion_unit_bypass == pytest.approx(0.2)
    assert cfg2.barrier_not_occupied_stp == pytest.approx(0.7 ** 2)
    # loop-only sampling disables TAD contacts (cli.cpp:970-983)
    assert api.make_config(contact_sampling_strategy=4).tad_to_loop_contact_ratio == 0.0
    assert np.isinf(api.make_config(contact_sampling_strategy=2 | 1).tad_to_loop_contact_ratio)


def test_barrier_stp_math():
    # reference: test/units/simulation_internal/extrusion_barriers_test.cpp:36-97
    L = _lib.lib()
    assert L.modle_hip_occupancy_from_stp(1.0, 0.0) == 1.0
    assert L.modle_hip_occupancy_from_stp(0.0, 1.0) == 0.0
    assert L.modle_hip_occupancy_from_stp(0.7, 0.7) == 0.5
    assert L.modle_hip_occupancy_from_stp(0.7, 0.5) == pytest.approx(0.625, rel=1e-12)
    assert L.modle_hip_occupancy_from_stp(1.0, 0.5) == 1.0
    assert L.modle_hip_occupancy_from_stp(0.7, 1.0) == 0.0
    for stp_inactive, occ in ((0.7, 0.85), (0.65, 0.93)):
        stp_active = L.modle_hip_stp_active_from_occupancy(stp_inactive, occ)
        assert L.modle_hip_occupancy_from_stp(stp_active, stp_inactive) == pytest.approx(occ)


def test_interval_hash_matches_xxhash():
    for name, size, start, end, seed in (("chr1", 248956422, 0, 248956422, 0),
                                         ("chrY", 57227415, 0, 57227415, 0),
                                         ("chrUn_KI270302v1_a_rather_long_name", 2274, 10, 2000, 7),
                                         ("x" * 150, 12345, 1, 12345, 2 ** 63 + 5)):
        data = name.encode() + b"".join(int(v).to_bytes(8, "little") for v in (size, start, end))
        assert api.interval_hash(name, size, start, end, seed) == \
            xxhash.xxh3_64(data, seed=seed).intdigest()
    # values quoted in SURVEY.md section 8c
    assert api.interval_hash("chr1", 248956422, 0, 248956422, 0) == 0xA53D8E35875B84B9


def test_prng_seed_and_jump_match_oracle(oracle):
    assert api.prng_seed(752741483) == [0x2A3BC28B8FC13C5A, 0xDB997EC403E6D05D,
                                        0x3B9841261CC6FECA, 0x3943D8B92B198BDB]
    g = oracle.prng_from_seed(10556020843759504871)
    assert api.prng_seed(10556020843759504871) == g.state()
    oracle.lib().mo_prng_jump(C.byref(g))
    assert api.prng_jump(api.prng_seed(10556020843759504871)) == g.state()


def test_make_tasks_matches_oracle(oracle):
    cfg = api.make_config(num_cells=37, seed=11)
    mine = api.make_tasks(cfg, "chr7", 159345973, 0, 159345973, first_task_id=100)
    ref = oracle.make_tasks(cfg, "chr7", 159345973, 0, 159345973, first_id=100)
    assert len(mine) == len(ref) == 37
    total = 0
    for a, b in zip(mine, ref):
        assert (a.id, a.cell_id, a.num_lefs, a.num_target_contacts, a.num_target_epochs) == \
               (b.id, b.cell_id, b.num_lefs, b.num_target_contacts, b.num_target_epochs)
        assert list(a.prng) == list(b.prng)
        total += a.num_target_contacts
    nrows, ncols = api.matrix_shape(cfg, 159345973)
    assert total == nrows * ncols  # density 1.0: the split never overshoots
    assert api.compute_num_lefs(cfg, 248956422) == 4979
    assert api.compute_contacts_per_epoch(cfg, 4979) == 797


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.ModleHipError, match="no CPU fallback"):
        api.Simulator(api.make_config())


def test_device_division_by_a_uniform_bucket_is_exact():
    """phase_bind divides the raw PRNG output by the (wave-uniform) bucket of uniform_int with a
    double-precision product and a remainder fix-up instead of a 64-bit division (sim_rng.h:
    udiv_by_uniform); the quotient must be the integer quotient for every input, including the
    multiples of the bucket and their neighbours."""
    import ctypes as C

    from phase_backend import emu_lib

    lib = emu_lib()
    lib.emu_check_udiv_by_uniform.restype = C.c_uint64
    lib.emu_check_udiv_by_uniform.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(3)
    n = 400_000
    raws = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    ranges = rng.integers(1, 2**32 - 1, size=n, dtype=np.uint64)
    ranges[:64] = (np.uint64(1) << (np.arange(64, dtype=np.uint64) % np.uint64(32))) + np.arange(64, dtype=np.uint64) // np.uint64(32)
    assert lib.emu_check_udiv_by_uniform(raws.ctypes.data, ranges.ctypes.data, n) == 0


def test_c11_consumer_compiles_and_agrees_with_the_python_view(tmp_path):
    """include/modle_hip.h and include/modle_cooler.h are valid C11 for a real consumer (not only
    for the regex above): a C program compiled with -Wall -Wextra -Werror links against the two
    libraries and its host-logic calls give what the ctypes view gives."""
    import shutil
    import subprocess

    from modle_amd import cooler

    cooler.lib()
    src = os.path.join(ROOT, "tests", "c_consumer", "consumer.c")
    exe = str(tmp_path / "consumer")
    libdir = os.path.join(ROOT, "modle_amd")
    cc = shutil.which("gcc") or shutil.which("cc")
    assert cc, "no C compiler"
    subprocess.run([cc, "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic",
                    "-I" + os.path.join(ROOT, "include"), src, "-o", exe, "-L" + libdir,
                    "-lmodle_hip", "-lmodle_cooler", "-Wl,-rpath," + libdir,
                    "-Wl,-rpath,/opt/rocm/lib"],
                   check=True, capture_output=True, text=True)
    cool, bw = str(tmp_path / "c.cool"), str(tmp_path / "c.bw")
    out = subprocess.run([exe, cool, bw], check=True, capture_output=True, text=True).stdout.split("\n")
    cfg = api.make_config(num_cells=4)
    size = 5_000_000
    tasks = api.make_tasks(cfg, "chrC", size, 0, size)
    assert out[0] == f"hash {api.interval_hash('chrC', size, 0, size, cfg.seed)}"
    assert out[1] == "shape %d %d" % api.matrix_shape(cfg, size)
    assert out[2] == f"nlefs {api.compute_num_lefs(cfg, size)}"
    for i in range(4):
        t = tasks[i]
        assert out[3 + i] == f"task {t.cell_id} {t.num_target_contacts} {t.prng[0]} {t.prng[3]}"
    assert out[7] == "sorted 100 500 900 2 0.8 0.1"
    assert float(out[8].split()[1]) == api.stp_active_from_occupancy(cfg.barrier_not_occupied_stp, 0.85)
    assert out[9] in ("create ok", "create refused")
    assert out[10] == "genome 1 50000 110 2"  # midpoint (100 + 120 + 1) / 2, strand '+' => DIR_REV
    assert out[11] == "cooler ok" and os.path.getsize(cool) > 0
    assert out[12] == "bigwig ok" and os.path.getsize(bw) > 0


def test_sort_barriers_is_stable_and_matches_the_oracle(oracle):
    import ctypes as C

    rng = np.random.default_rng(5)
    n = 500
    pos = rng.integers(0, 200, size=n).astype(np.uint64)  # many duplicates
    dirs = rng.integers(1, 3, size=n).astype(np.uint8)
    sa, si = rng.random(n), rng.random(n)
    p2, d2, a2, i2 = api.sort_barriers(pos, dirs, sa, si)
    order = np.argsort(pos, kind="stable")
    assert np.array_equal(p2, pos[order]) and np.array_equal(d2, dirs[order])
    assert np.array_equal(a2, sa[order]) and np.array_equal(i2, si[order])
    L = oracle.lib()
    u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
    u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
    f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    L.mo_sort_barriers.argtypes = [C.c_size_t, u64p, u8p, f64p, f64p]
    L.mo_sort_barriers.restype = None
    p3, d3, a3, i3 = pos.copy(), dirs.copy(), sa.copy(), si.copy()
    L.mo_sort_barriers(n, p3, d3, a3, i3)
    assert np.array_equal(p2, p3) and np.array_equal(d2, d3)
    assert np.array_equal(a2, a3) and np.array_equal(i2, i3)

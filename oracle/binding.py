"""ctypes binding of oracle/libmodle_oracle.so (test infrastructure, see modle_oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

from modle_amd.params import CellResult, Config, Task

_HERE = os.path.dirname(os.path.abspath(__file__))
# MODLE_ORACLE_LIB=libmodle_oracle_libm.so selects the build that calls glibc's log / exp / pow
# (make -C oracle libmodle_oracle_libm.so; tools/libm_flip_rate.py): a measuring instrument, never
# the parity oracle
_SO = os.path.join(_HERE, os.environ.get("MODLE_ORACLE_LIB", "libmodle_oracle.so"))

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


class Prng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4), ("count", C.c_uint64)]

    def state(self):
        return [int(x) for x in self.s]


def build():
    """(Re)build the oracle shared library with the committed Makefile."""
    subprocess.run(["make", "-C", _HERE, os.path.basename(_SO)], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    P = C.POINTER
    L.mo_prng_seed.argtypes = [P(Prng), C.c_uint64]
    L.mo_prng_next.argtypes = [P(Prng)]
    L.mo_prng_next.restype = C.c_uint64
    L.mo_prng_jump.argtypes = [P(Prng)]
    L.mo_set_rng_policy.argtypes = [C.c_int]
    L.mo_set_rng_policy.restype = None
    L.mo_get_rng_policy.restype = C.c_int
    L.mo_bernoulli.argtypes = [P(Prng), C.c_double]
    L.mo_bernoulli.restype = C.c_int
    for name in ("mo_canonical", "mo_uniform_01"):
        getattr(L, name).argtypes = [P(Prng)]
        getattr(L, name).restype = C.c_double
    L.mo_uniform_int.argtypes = [P(Prng), C.c_uint64, C.c_uint64]
    L.mo_uniform_int.restype = C.c_uint64
    L.mo_normal.argtypes = [P(Prng), C.c_double, C.c_double]
    L.mo_normal.restype = C.c_double
    L.mo_poisson.argtypes = [P(Prng), C.c_double]
    L.mo_poisson.restype = C.c_uint64
    L.mo_binomial.argtypes = [P(Prng), C.c_int64, C.c_double]
    L.mo_binomial.restype = C.c_int64
    L.mo_genextreme.argtypes = [P(Prng), C.c_double, C.c_double, C.c_double]
    L.mo_genextreme.restype = C.c_double
    L.mo_xxh3_64.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
    L.mo_xxh3_64.restype = C.c_uint64
    L.mo_interval_hash.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    L.mo_interval_hash.restype = C.c_uint64
    L.mo_compute_num_lefs.argtypes = [P(Config), C.c_uint64]
    L.mo_compute_num_lefs.restype = C.c_uint64
    L.mo_compute_contacts_per_epoch.argtypes = [P(Config), C.c_uint64]
    L.mo_compute_contacts_per_epoch.restype = C.c_uint64
    L.mo_matrix_shape.argtypes = [P(Config), C.c_uint64, P(C.c_uint64), P(C.c_uint64)]
    L.mo_make_tasks.argtypes = [P(Config), C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                C.c_uint64, P(Task)]
    L.mo_stp_active_from_occupancy.argtypes = [C.c_double, C.c_double]
    L.mo_stp_active_from_occupancy.restype = C.c_double
    L.mo_occupancy_from_stp.argtypes = [C.c_double, C.c_double]
    L.mo_occupancy_from_stp.restype = C.c_double

    sz = C.c_size_t
    L.mo_rank_lefs.argtypes = [sz, u64p, u64p, u64p, u64p, u64p, C.c_int]
    L.mo_adjust_moves.argtypes = [C.c_uint64, C.c_uint64, sz] + [u64p] * 7
    L.mo_clamp_moves.argtypes = [C.c_uint64, C.c_uint64, sz] + [u64p] * 5
    L.mo_detect_units_at_interval_boundaries.argtypes = (
        [C.c_uint64, C.c_uint64, sz] + [u64p] * 9 + [P(C.c_uint64), P(C.c_uint64)])
    L.mo_detect_lef_bar_collisions.argtypes = (
        [P(Config), sz] + [u64p] * 7 + [sz, u64p, u8p, u8p, u64p, u64p, P(Prng), C.c_uint64,
                                        C.c_uint64])
    L.mo_detect_primary_lef_lef_collisions.argtypes = (
        [P(Config), sz] + [u64p] * 6 + [sz] + [u64p] * 3 + [P(Prng), C.c_uint64, C.c_uint64])
    L.mo_correct_moves_for_lef_bar_collisions.argtypes = [sz] + [u64p] * 7
    L.mo_correct_moves_for_primary_lef_lef_collisions.argtypes = [sz] + [u64p] * 8
    L.mo_process_secondary_lef_lef_collisions.argtypes = (
        [P(Config), sz] + [u64p] * 8 + [P(Prng), C.c_uint64, C.c_uint64])
    L.mo_fix_secondary_lef_lef_collisions.argtypes = (
        [C.c_uint64, C.c_uint64, sz] + [u64p] * 8 + [C.c_uint64, C.c_uint64])
    L.mo_process_collisions.argtypes = (
        [P(Config), C.c_uint64, C.c_uint64, sz] + [u64p] * 7 + [sz, u64p, u8p, u8p, u64p, u64p,
                                                                P(Prng), C.c_int])
    L.mo_generate_moves.argtypes = (
        [P(Config), C.c_uint64, C.c_uint64, sz] + [u64p] * 7 + [C.c_int, P(Prng), C.c_int])
    L.mo_simulate_cell.argtypes = [P(Config), C.c_uint64, C.c_uint64, sz, u64p, u8p, f64p, f64p,
                                   P(Task), u32p, C.c_uint64, C.c_uint64, P(C.c_uint64),
                                   C.c_void_p, P(CellResult)]
    L.mo_simulate_cell.restype = C.c_int
    L.mo_simulate_interval.argtypes = [P(Config), C.c_uint64, C.c_uint64, sz, u64p, u8p, f64p,
                                       f64p, P(Task), sz, u32p, C.c_uint64, C.c_uint64,
                                       P(C.c_uint64), C.c_void_p, P(CellResult), C.c_int]
    L.mo_simulate_interval.restype = C.c_int
    _lib = L
    return L


def prng_from_seed(seed):
    g = Prng()
    lib().mo_prng_seed(C.byref(g), seed)
    return g


def prng_from_state(state):
    g = Prng()
    for i in range(4):
        g.s[i] = int(state[i])
    g.count = 0
    return g


def make_tasks(cfg, name, chrom_size, start, end, first_id=0):
    tasks = (Task * int(cfg.num_cells))()
    lib().mo_make_tasks(C.byref(cfg), name.encode(), chrom_size, start, end, first_id, tasks)
    return tasks


def matrix_shape(cfg, size_bp):
    nr, nc = C.c_uint64(), C.c_uint64()
    lib().mo_matrix_shape(C.byref(cfg), size_bp, C.byref(nr), C.byref(nc))
    return nr.value, nc.value


def simulate_interval(cfg, start, end, bar_pos, bar_dir, stp_active, stp_inactive, tasks,
                      nthreads=1, track_occupancy=True):
    """Runs every task of one interval; returns (contacts, missed, occupancy, results)."""
    L = lib()
    nrows, ncols = matrix_shape(cfg, end - start)
    contacts = np.zeros(nrows * ncols + 1, dtype=np.uint32)
    occ = np.zeros(ncols, dtype=np.uint64) if track_occupancy else None
    missed = C.c_uint64(0)
    n = len(tasks)
    results = (CellResult * n)()
    bar_pos = np.ascontiguousarray(bar_pos, dtype=np.uint64)
    bar_dir = np.ascontiguousarray(bar_dir, dtype=np.uint8)
    stp_active = np.ascontiguousarray(stp_active, dtype=np.float64)
    stp_inactive = np.ascontiguousarray(stp_inactive, dtype=np.float64)
    rc = L.mo_simulate_interval(C.byref(cfg), start, end, len(bar_pos), bar_pos, bar_dir,
                                stp_active, stp_inactive, tasks, n, contacts, nrows, ncols,
                                C.byref(missed), occ.ctypes.data if occ is not None else None,
                                results, nthreads)
    if rc != 0:
        raise RuntimeError(f"oracle failed with code {rc}")
    return contacts, missed.value, occ, results


class rng_policy:
    """context manager: run the oracle with the PHILOX generator policy (see modle_oracle.c)"""

    def __init__(self, philox=True):
        self.philox = philox

    def __enter__(self):
        self.prev = lib().mo_get_rng_policy()
        lib().mo_set_rng_policy(1 if self.philox else 0)

    def __exit__(self, *exc):
        lib().mo_set_rng_policy(self.prev)
        return False

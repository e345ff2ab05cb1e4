"""The reference's small unit tests that touch the hot path (tests/golden/reference_unit_vectors.json)
on the CPU oracle and on the device code under the lane emulator; the property tests also compare
the two word for word (positions, moves, ranks, PRNG outputs consumed)."""
import ctypes as C

import numpy as np
import pytest

from unit_vector_runner import DeviceUnits, OracleUnits, all_vectors, run_vector

VECTORS = all_vectors()
IDS = [v["name"] for _, v in VECTORS]


def _emu_backend(oracle):
    from modle_amd.params import Config
    from phase_backend import emu_lib, emu_phases, emu_size_class

    u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
    libs = {}

    def lib(variant):
        if variant not in libs:
            L = emu_lib(variant)
            L.emu_test_units.argtypes = [C.POINTER(Config), C.c_uint32, u64p, C.c_size_t, C.c_uint64,
                                         C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), u64p]
            L.emu_test_units.restype = C.c_int
            libs[variant] = L
        return libs[variant]

    def units(cfg, what, pairs, nrows, ncols, contacts, missed):
        pairs = np.ascontiguousarray(pairs, dtype=np.uint64).reshape(-1)
        out = np.zeros(len(pairs), dtype=np.uint64)
        m = C.c_uint64(missed)
        # (65 536 elements or more: the WIDE size class, like the product's modle_hip_test_units)
        L = lib("wide" if emu_size_class(cfg, len(pairs) // 2) != 0 else None)
        rc = L.emu_test_units(C.byref(cfg), what, pairs, len(pairs) // 2, nrows, ncols,
                              contacts.ctypes.data if contacts is not None else None, C.byref(m), out)
        assert rc == 0
        return out, m.value

    return DeviceUnits("emulator", units, emu_phases, oracle)


@pytest.mark.parametrize("group,v", VECTORS, ids=IDS)
def test_reference_unit_vector_on_oracle(oracle, group, v):
    run_vector(OracleUnits(oracle), group, v)


@pytest.mark.parametrize("group,v", VECTORS, ids=IDS)
def test_reference_unit_vector_on_emulated_device_code(oracle, group, v):
    be = _emu_backend(oracle)
    kw = {"iters": 40} if v["name"].startswith("Generate LEF moves") else {}
    got = run_vector(be, group, v, **kw)
    if group == "property_tests":
        st_o, n_o = run_vector(OracleUnits(oracle), group, v, **kw)
        st_d, n_d = got
        assert n_o == n_d, "PRNG outputs consumed"
        fields = ["rev_pos", "fwd_pos", "epoch", "rev_rank", "fwd_rank"]
        if not v["name"].startswith("Bind"):  # the move arrays mean nothing before generate_moves
            fields += ["rev_moves", "fwd_moves"]
        for f in fields:
            assert np.array_equal(getattr(st_o, f), getattr(st_d, f)), f


def test_loop_stats_bit_identical_on_random_loops(oracle):
    """beyond the reference's one vector: oracle and device code agree to the last bit on the
    burn-in statistics of random loop-size sets (sequential fp64 accumulation)"""
    rng = np.random.default_rng(9)
    be_o, be_d = OracleUnits(oracle), _emu_backend(oracle)
    for n in (1, 2, 63, 64, 65, 300, 511, 512, 513, 1000, 4979, 70000):  # (> 65536: the scattering form)
        rev = rng.integers(1, 200_000_000, size=n).astype(np.uint64)
        fwd = rev + rng.integers(0, 3_000_000, size=n).astype(np.uint64)
        a, b = be_o.loop_stats(rev, fwd), be_d.loop_stats(rev, fwd)
        assert a["mean"] == b["mean"] and a["std"] == b["std"], n
    # shapes that are hard for the device's fold (it replaces the chain of dependent additions by exact
    # prefix sums between the additions that tie or cross a binade, sim_burnin.h: fold_terms_exact):
    # tiny integers with a dyadic mean (every rounding a tie or exact), all equal (running sum zero),
    # two values, rare huge terms, powers of two
    for shape, n in [(s, n) for s in range(6) for n in (64, 1000, 4096, 5000)]:
        rev = rng.integers(1, 200_000_000, size=n).astype(np.uint64)
        size = [rng.integers(0, 4, size=n), np.full(n, 12345), rng.integers(0, 2, size=n) * 1_000_000,
                np.where(rng.integers(0, 64, size=n) == 0, 4_000_000_000, rng.integers(0, 100, size=n)),
                1 << rng.integers(0, 31, size=n), np.where(rng.integers(0, 2, size=n) == 0, 0,
                                                            rng.integers(0, 200_000_000, size=n))][shape]
        fwd = rev + size.astype(np.uint64)
        a, b = be_o.loop_stats(rev, fwd), be_d.loop_stats(rev, fwd)
        assert a["mean"] == b["mean"] and a["std"] == b["std"], (shape, n)

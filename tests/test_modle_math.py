"""The software log / exp / pow shared by the device code and the oracle
(modle_amd/csrc/modle_math.h; SURVEY.md H5 "libm differences").

* accuracy: below one ulp against mpmath over the argument ranges the path uses and beyond;
* special values;
* parity by construction: the device code under the lane emulator (here) and on the GPU
  (tests/test_gpu_unit_vectors.py::test_math_bit_identical_on_gpu) returns the same BITS as the
  oracle for every argument -- they compile the same source with contraction off."""
import ctypes as C

import mpmath as mp
import numpy as np
import pytest

mp.mp.prec = 200


def math_lib(oracle):
    L = oracle.lib()
    for f in ("mo_math_log", "mo_math_exp"):
        getattr(L, f).argtypes = [C.c_double]
        getattr(L, f).restype = C.c_double
    L.mo_math_pow.argtypes = [C.c_double, C.c_double]
    L.mo_math_pow.restype = C.c_double
    return L


def ulp_error(got, exact):
    if exact == 0:
        return 0.0 if got == 0 else float("inf")
    e = int(mp.floor(mp.log(abs(exact), 2)))
    return float(abs(mp.mpf(got) - exact) / mp.ldexp(1, max(e - 52, -1074)))


def path_arguments(rng, n):
    """(x, y) pairs shaped like the call sites: log(u) and pow(-log u, xi) of the GEV noise,
    exp(-x^2/2) of the ziggurat wedges, exp(-mean) and the log arguments of PTRD / BTRD,
    pow(1 - p, t) of the binomial inversion"""
    u = rng.random(n)
    u[u == 0] = 0.5
    x = np.concatenate([u, -np.log(u), rng.random(n) * 40 + 1e-3, 1 + rng.normal(0, 1e-4, n),
                        rng.random(n) * 0.5 + 0.5, np.exp(rng.uniform(-300, 300, n))])
    y = np.concatenate([-(rng.random(n) * 4) ** 2 / 2, rng.normal(0, 0.01, n), -rng.random(n) * 10,
                        rng.uniform(-30, 30, n), rng.integers(1, 4000, n).astype(float),
                        rng.normal(0, 2, n)])
    return x.astype(np.float64), y.astype(np.float64)


def test_accuracy_below_one_ulp(oracle):
    L = math_lib(oracle)
    rng = np.random.default_rng(1)
    x, y = path_arguments(rng, 1500)
    worst = {"log": 0.0, "exp": 0.0, "pow": 0.0}
    for a, b in zip(x, y):
        a, b = float(a), float(b)
        worst["log"] = max(worst["log"], ulp_error(L.mo_math_log(a), mp.log(mp.mpf(a))))
        if -745.0 < b < 709.0:
            worst["exp"] = max(worst["exp"], ulp_error(L.mo_math_exp(b), mp.exp(mp.mpf(b))))
        else:
            assert L.mo_math_exp(b) == (float("inf") if b > 0 else 0.0)
        exact = mp.power(mp.mpf(a), mp.mpf(b))
        if mp.mpf(1e-300) < exact < mp.mpf(1e300):
            worst["pow"] = max(worst["pow"], ulp_error(L.mo_math_pow(a, b), exact))
    assert worst["log"] < 0.51 and worst["exp"] < 0.75 and worst["pow"] < 0.75, worst


def test_special_values(oracle):
    L = math_lib(oracle)
    inf = float("inf")
    assert L.mo_math_log(1.0) == 0.0 and L.mo_math_log(0.0) == -inf and L.mo_math_log(inf) == inf
    assert np.isnan(L.mo_math_log(-1.0)) and np.isnan(L.mo_math_log(float("nan")))
    assert L.mo_math_log(5e-324) == pytest.approx(-744.4400719213812, rel=1e-15)
    assert L.mo_math_exp(0.0) == 1.0 and L.mo_math_exp(1000.0) == inf and L.mo_math_exp(-1000.0) == 0.0
    assert L.mo_math_exp(-745.0) == 5e-324 and L.mo_math_exp(709.7) == pytest.approx(1.6549840276802644e308, rel=1e-15)
    assert L.mo_math_pow(2.0, 10.0) == 1024.0 and L.mo_math_pow(-8.0, 3.0) == -512.0
    assert L.mo_math_pow(7.3, 0.0) == 1.0 and L.mo_math_pow(1.0, 1e300) == 1.0
    assert np.isnan(L.mo_math_pow(-8.0, 0.5)) and L.mo_math_pow(0.0, -1.0) == inf
    assert L.mo_math_pow(2.0, 1024.0) == inf and L.mo_math_pow(2.0, -1080.0) == 0.0
    assert L.mo_math_pow(0.5, inf) == 0.0 and L.mo_math_pow(0.5, -inf) == inf


def oracle_bits(L, x, y):
    lg = np.array([L.mo_math_log(float(a)) for a in x]).view(np.uint64)
    ex = np.array([L.mo_math_exp(float(b)) for b in y]).view(np.uint64)
    pw = np.array([L.mo_math_pow(float(a), float(b)) for a, b in zip(x, y)]).view(np.uint64)
    return lg, ex, pw


def test_emulated_device_code_returns_the_same_bits(oracle):
    from test_oracle_unit_vectors import _emu_backend
    from unit_vector_runner import base_config

    L = math_lib(oracle)
    x, y = path_arguments(np.random.default_rng(2), 400)
    lg, ex, pw = oracle_bits(L, x, y)
    be = _emu_backend(oracle)
    pairs = np.stack([x.view(np.uint64), y.view(np.uint64)], axis=1)
    out, _ = be._units(base_config(), 4, pairs, 0, 0, None, 0)
    assert np.array_equal(out[0::2], lg) and np.array_equal(out[1::2], ex)
    out, _ = be._units(base_config(), 5, pairs, 0, 0, None, 0)
    assert np.array_equal(out[0::2], pw)
    assert np.array_equal(out[1::2], np.sqrt(x).view(np.uint64))

"""Statistical validator (modle_amd/evaluate.py; reference: src/modle_tools/eval.cpp:328-497,
src/stats/correlation_impl.hpp): stripe extraction from the band layout, Pearson / Spearman per
stripe against scipy, the reference's special cases, and the cooler read-back it works on."""
import numpy as np
import pytest
from scipy import stats

from modle_amd import cooler, evaluate


def _dense_from_band(band, nrows, ncols):
    m = np.zeros((ncols, ncols))
    for j in range(ncols):
        for i in range(min(nrows, j + 1)):
            m[j - i, j] = m[j, j - i] = band[j * nrows + i]
    return m


def _random_band(rng, nrows, ncols, lam):
    band = rng.poisson(lam, size=nrows * ncols + 1).astype(np.uint32)
    for j in range(min(nrows, ncols)):
        band[j * nrows + j + 1:(j + 1) * nrows] = 0
    return band


def test_stripes_follow_the_reference_orientation():
    rng = np.random.default_rng(0)
    nrows, ncols = 5, 12
    band = _random_band(rng, nrows, ncols, 20)
    m = _dense_from_band(band, nrows, ncols)
    v = evaluate.stripes(evaluate.band_to_2d(band, nrows, ncols), "vertical")
    h = evaluate.stripes(evaluate.band_to_2d(band, nrows, ncols), "horizontal")
    for c in range(ncols):
        # unsafe_get_column: from the diagonal towards the periphery, zero padded
        assert v[c].tolist() == [m[c - i, c] if c - i >= 0 else 0.0 for i in range(nrows)]
        assert h[c].tolist() == [m[c, c + i] if c + i < ncols else 0.0 for i in range(nrows)]


@pytest.mark.parametrize("direction", ["vertical", "horizontal"])
def test_pearson_and_spearman_match_scipy(direction):
    rng = np.random.default_rng(1)
    nrows, ncols = 40, 90
    a = _random_band(rng, nrows, ncols, 6)   # many ties: exercises the average ranks
    b = (a + rng.poisson(3, size=a.shape)).astype(np.uint32)
    sa = evaluate.stripes(evaluate.band_to_2d(a, nrows, ncols), direction)
    sb = evaluate.stripes(evaluate.band_to_2d(b, nrows, ncols), direction)
    for metric, fn in (("pearson", stats.pearsonr), ("spearman", stats.spearmanr)):
        val, pv = evaluate.compare(a, b, nrows, ncols, metric, direction)
        for c in range(5, ncols - 5):
            r, p = fn(sa[c], sb[c])
            assert val[c] == pytest.approx(r, abs=1e-12), (metric, c)
            assert pv[c] == pytest.approx(p, rel=1e-6, abs=1e-300), (metric, c)


def test_special_cases_and_masking():
    nrows, ncols = 6, 10
    a = np.zeros(nrows * ncols + 1, dtype=np.uint32)
    b = _random_band(np.random.default_rng(2), nrows, ncols, 5)
    rho, _ = evaluate.compare(a, b, nrows, ncols, "spearman")
    assert np.all(rho == 1.0)  # a stripe without contacts on one side scores 1 (correlation_impl.hpp)
    pcc, _ = evaluate.compare(a, b, nrows, ncols, "pearson")
    assert np.all(np.isnan(pcc))
    # identical matrices
    for metric, want in (("pearson", 1.0), ("spearman", 1.0), ("rmse", 0.0), ("eucl_dist", 0.0)):
        val, _ = evaluate.compare(b, b, nrows, ncols, metric)
        ok = ~np.isnan(val)
        assert ok.sum() >= ncols - 2 and np.allclose(val[ok], want)
    # masking: pixels that are zero in either matrix do not count
    c = b.copy()
    c[7 * nrows + 2] = 0
    d = b.copy()
    d[7 * nrows + 2] += 1000
    m1, pv = evaluate.compare(b, c, nrows, ncols, "rmse", mask_zero_pixels=True)
    assert m1[7] == 0.0 and np.isnan(pv[7])
    m2, _ = evaluate.compare(b, d, nrows, ncols, "rmse", mask_zero_pixels=True)
    assert m2[7] > 100


def test_cooler_read_back_and_file_comparison(tmp_path):
    rng = np.random.default_rng(3)
    chroms = [("chrA", 400_000), ("chrB", 250_000)]
    bin_size, nrows = 5000, 20
    bands = {}
    paths = []
    for k in range(2):
        path = str(tmp_path / f"m{k}.cool")
        with cooler.CoolerWriter(path, chroms, bin_size) as w:
            for name, size in chroms:
                ncols = -(-size // bin_size)
                band = _random_band(rng, nrows, ncols, 8)
                bands[(k, name)] = (band, ncols)
                w.append(name, band, nrows, ncols)
        paths.append(path)
    for name, size in chroms:
        got, nr, nc, bs, missed = evaluate.read_cooler_band(paths[0], name, nrows * bin_size)
        band, ncols = bands[(0, name)]
        assert (nr, nc, bs, missed) == (nrows, ncols, bin_size, 0)
        assert np.array_equal(got[:nr * nc], band[:nr * nc])
        # a narrower band: the contacts beyond it are reported, not dropped silently
        got, nr, nc, bs, missed = evaluate.read_cooler_band(paths[0], name, 5 * bin_size)
        b2 = evaluate.band_to_2d(band, nrows, ncols)
        assert nr == 5 and missed == int(b2[:, 5:].sum())
        assert np.array_equal(evaluate.band_to_2d(got, 5, ncols), b2[:, :5])
    res = evaluate.compare_coolers(paths[0], paths[0], ["chrA", "chrB"], nrows * bin_size)
    assert res["chrA"]["vertical"]["median"] == pytest.approx(1.0)
    res = evaluate.compare_coolers(paths[0], paths[1], ["chrA"], nrows * bin_size, "spearman")
    assert abs(res["chrA"]["horizontal"]["mean"]) < 0.3  # independent noise: no correlation
    with pytest.raises(cooler.CoolerError):
        evaluate.read_cooler_band(paths[0], "chrMissing", 100000)

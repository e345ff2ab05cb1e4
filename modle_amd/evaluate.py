"""Statistical validator: per-stripe comparison of two contact matrices, the method of the
reference's `modle_tools evaluate` (reference:
src/modle_tools/eval.cpp:328-497, src/stats/correlation_impl.hpp).

For every bin of a chromosome the pixels of its VERTICAL stripe (bin c against the bins upstream,
from the diagonal outwards: (c, c), (c-1, c), ...) and of its HORIZONTAL stripe ((r, r), (r, r+1),
...), `nrows` pixels each (band width), are compared between a reference and a target matrix with
Pearson's or Spearman's correlation (ties get average ranks; a stripe that is all zeros on either
side scores 1 for Spearman, like the reference), RMSE or the Euclidean distance.  Optional masking
of pixels that are zero in either matrix turns the metric into its weighted form (weight 0 / 1).

Used here to qualify outputs that are NOT expected to be bit-identical to the exact mode: the
PHILOX generator policy and deliberate tie-break deviations.  It is never part of a parity claim.
"""
import ctypes

import numpy as np

from . import cooler as _cooler

METRICS = ("pearson", "spearman", "rmse", "eucl_dist")


def band_to_2d(band, nrows, ncols):
    """band[j * nrows + i] (i = |row - col|, j = max(row, col)) as an (ncols, nrows) array"""
    return np.asarray(band[:nrows * ncols], dtype=np.float64).reshape(ncols, nrows)


def stripes(band2d, direction):
    """(ncols, nrows) array whose row c is the stripe of bin c, diagonal first, zero padded"""
    ncols, nrows = band2d.shape
    if direction == "vertical":
        out = band2d.copy()
        i = np.arange(nrows)[None, :]
        out[i > np.arange(ncols)[:, None]] = 0.0  # pixels above the first row do not exist
        return out
    if direction != "horizontal":
        raise ValueError(direction)
    out = np.zeros_like(band2d)
    for i in range(nrows):  # pixel (r, r + i) is stored at band2d[r + i, i]
        out[:ncols - i, i] = band2d[i:, i]
    return out


def _rank_rows(x):
    """average ranks along axis 1 (ties averaged), vectorised over the rows"""
    order = np.argsort(x, axis=1, kind="stable")
    sx = np.take_along_axis(x, order, axis=1)
    n = x.shape[1]
    ranks = np.empty_like(x)
    base = np.arange(n, dtype=np.float64)
    for r in range(x.shape[0]):  # runs of equal values share the mean of their positions
        row = sx[r]
        starts = np.flatnonzero(np.r_[True, row[1:] != row[:-1]])
        ends = np.r_[starts[1:], n]
        avg = np.repeat((starts + ends - 1) / 2.0, ends - starts)
        ranks[r, order[r]] = avg
    return ranks


def _weighted_pcc(a, b, w):
    sw = w.sum(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        ma = (a * w).sum(axis=1) / sw
        mb = (b * w).sum(axis=1) / sw
        da, db = a - ma[:, None], b - mb[:, None]
        cov = (w * da * db).sum(axis=1)
        va = (w * da * da).sum(axis=1)
        vb = (w * db * db).sum(axis=1)
        return np.clip(cov / np.sqrt(va * vb), -1.0, 1.0)


def weighted_ranks(v, w):
    """compute_weighted_element_ranks (reference: src/stats/correlation_impl.hpp:265-324): the
    rank of an element is the sum of the weights of the elements up to and including it in sorted
    order; t tied elements share (weights before them) + (t + 1) / 2 * (their mean weight)"""
    v = np.asarray(v, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    order = np.argsort(v, kind="stable")
    sv, sw = v[order], w[order]
    n = len(v)
    starts = np.flatnonzero(np.r_[True, sv[1:] != sv[:-1]]) if n else np.zeros(0, dtype=np.int64)
    ends = np.r_[starts[1:], n]
    group_w = np.add.reduceat(sw, starts) if n else sw
    before = np.r_[0.0, np.cumsum(group_w)[:-1]] if n else sw
    size = (ends - starts).astype(np.float64)
    # (a single element: weights before it + its own weight = the same formula with t = 1)
    rank_of_group = before + (size + 1.0) / 2.0 * (group_w / size)
    ranks = np.empty(n, dtype=np.float64)
    ranks[order] = np.repeat(rank_of_group, ends - starts)
    return ranks


def pearson(v1, v2, weights=None):
    """(pcc, p-value) like stats::Pearson<> (reference: src/stats/correlation_impl.hpp:29-116):
    weighted form when `weights` is given (its p-value is NaN)"""
    a = np.asarray(v1, dtype=np.float64)[None, :]
    b = np.asarray(v2, dtype=np.float64)[None, :]
    w = np.ones_like(a) if weights is None else np.asarray(weights, dtype=np.float64)[None, :]
    pcc = float(_weighted_pcc(a, b, w)[0])
    if weights is not None or np.isnan(pcc):
        return pcc, float("nan")
    from scipy import stats

    ab = a.shape[1] / 2.0 - 1.0
    return pcc, float(2.0 * stats.beta.cdf(0.5 * (1.0 - abs(pcc)), ab, ab))


def spearman(v1, v2, weights=None):
    """(rho, p-value) like stats::Spearman<> (reference: src/stats/correlation_impl.hpp:118-205):
    1 when either vector is all zeros; average ranks for ties; weighted ranks with `weights`"""
    a = np.asarray(v1, dtype=np.float64)
    b = np.asarray(v2, dtype=np.float64)
    if not a.any() or not b.any():
        rho = 1.0
    elif weights is None:
        rho = float(_weighted_pcc(_rank_rows(a[None, :]), _rank_rows(b[None, :]), np.ones((1, len(a))))[0])
    else:
        w = np.asarray(weights, dtype=np.float64)
        rho = float(_weighted_pcc(weighted_ranks(a, w)[None, :], weighted_ranks(b, w)[None, :], w[None, :])[0])
    if weights is not None or np.isnan(rho):
        return rho, float("nan")
    from scipy import stats

    dof = len(a) - 2.0
    with np.errstate(divide="ignore"):
        t = rho * np.sqrt(dof / ((1.0 + rho) * (1.0 - rho)))
    return rho, float(2.0 * stats.t.sf(abs(t), dof))


def compare(ref_band, tgt_band, nrows, ncols, metric="pearson", direction="vertical",
            mask_zero_pixels=False):
    """per-bin metric between two band matrices of the same shape; returns (values, pvalues)
    (p-values: Pearson via the beta distribution, Spearman via Student's t, as the reference;
    NaN for weighted metrics and for the distances)"""
    if metric not in METRICS:
        raise ValueError(f"metric must be one of {METRICS}")
    a = stripes(band_to_2d(ref_band, nrows, ncols), direction)
    b = stripes(band_to_2d(tgt_band, nrows, ncols), direction)
    w = np.ones_like(a)
    if mask_zero_pixels:
        w[(a == 0) | (b == 0)] = 0.0
    pv = np.full(ncols, np.nan)
    if metric == "rmse":
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.sqrt((w * (a - b) ** 2).sum(axis=1) / w.sum(axis=1)), pv
    if metric == "eucl_dist":
        return np.sqrt((w * (a - b) ** 2).sum(axis=1)), pv
    if metric == "spearman":
        degenerate = (a == 0).all(axis=1) | (b == 0).all(axis=1)
        if mask_zero_pixels:
            # the reference's weighted ranks with weights 0 / 1 (eval.cpp:331-344, 449-453)
            ra = np.stack([weighted_ranks(a[r], w[r]) for r in range(a.shape[0])])
            rb = np.stack([weighted_ranks(b[r], w[r]) for r in range(b.shape[0])])
            val = _weighted_pcc(ra, rb, w)
        else:
            val = _weighted_pcc(_rank_rows(a), _rank_rows(b), w)
        val[degenerate] = 1.0
    else:
        val = _weighted_pcc(a, b, w)
    if not mask_zero_pixels and nrows > 2:
        from scipy import stats

        with np.errstate(invalid="ignore", divide="ignore"):
            if metric == "pearson":
                ab = nrows / 2.0 - 1.0
                pv = 2.0 * stats.beta.cdf(0.5 * (1.0 - np.abs(val)), ab, ab)
            else:
                dof = nrows - 2.0
                t = val * np.sqrt(dof / ((1.0 + val) * (1.0 - val)))
                pv = 2.0 * stats.t.sf(np.abs(t), dof)
        pv[np.isnan(val)] = np.nan
    return val, pv


def summarize(values):
    v = values[~np.isnan(values)]
    if len(v) == 0:
        return {"n": 0}
    return {"n": int(len(v)), "mean": float(v.mean()), "median": float(np.median(v)),
            "p05": float(np.percentile(v, 5)), "min": float(v.min())}


def read_cooler_band(path, chrom, diagonal_width):
    """(band, nrows, ncols, bin_size, contacts beyond the band) of one chromosome of a .cool"""
    lb = _cooler.lib()
    lb.modle_cool_read_band.restype = ctypes.c_int
    lb.modle_cool_read_band.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p,
                                        ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
                                        ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64),
                                        ctypes.c_char_p, ctypes.c_size_t]
    err = ctypes.create_string_buffer(512)
    ncols, bs, missed = ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_uint64()
    rc = lb.modle_cool_read_band(path.encode(), chrom.encode(), 1, None, 0, ctypes.byref(ncols),
                                 ctypes.byref(bs), None, err, len(err))
    if rc != 0:
        raise _cooler.CoolerError(rc, err.value.decode())
    nrows = min(-(-int(diagonal_width) // bs.value), ncols.value)
    band = np.zeros(nrows * ncols.value + 1, dtype=np.uint32)
    rc = lb.modle_cool_read_band(path.encode(), chrom.encode(), nrows, band.ctypes.data, band.size,
                                 ctypes.byref(ncols), ctypes.byref(bs), ctypes.byref(missed), err, len(err))
    if rc != 0:
        raise _cooler.CoolerError(rc, err.value.decode())
    return band, nrows, ncols.value, bs.value, missed.value


def compare_coolers(ref_path, tgt_path, chroms, diagonal_width, metric="pearson",
                    mask_zero_pixels=False):
    """{chrom: {direction: summary}} for two cooler files of the same genome and resolution"""
    out = {}
    for chrom in chroms:
        a, nr, nc, bs, _ = read_cooler_band(ref_path, chrom, diagonal_width)
        b, nr2, nc2, bs2, _ = read_cooler_band(tgt_path, chrom, diagonal_width)
        if (nr, nc, bs) != (nr2, nc2, bs2):
            raise ValueError(f"{chrom}: the two files differ in shape or resolution")
        out[chrom] = {d: summarize(compare(a, b, nr, nc, metric, d, mask_zero_pixels)[0])
                      for d in ("vertical", "horizontal")}
    return out

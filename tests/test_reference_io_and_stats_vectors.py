"""Vectors the reference's own unit tests hold for the rows next to the hot path (SURVEY.md
section 8(f)), replayed through this repo's implementations.  Data only: extracted by
tests/golden/extract_reference_unit_vectors.py into tests/golden/reference_unit_vectors.json.

    test/units/stats/correlation_test.cpp:124-222       -> modle_amd/evaluate.py (row f4)
    test/units/libmodle_io/bed_parser_test.cpp:71-122   -> modle_genome_import  (row f2)
    test/units/libmodle_io/bigwig_test.cpp:100-135      -> modle_bw_* + tests/bigwig_reader.py (row f3)
"""
import json
import math
import os

import numpy as np
import pytest

from bigwig_reader import BigWig
from modle_amd import api, bigwig, evaluate, genome, params

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_unit_vectors.json")) as fh:
    VEC = json.load(fh)


@pytest.mark.parametrize("case", VEC["correlation"], ids=lambda c: c["name"])
def test_correlation_vectors(case):
    fn = evaluate.spearman if case["method"] == "spearman" else evaluate.pearson
    val, pv = fn(case["v1"], case["v2"], case["weights"])
    tol = case["rel_tolerance"]  # Catch2 WithinRel(DEFAULT_FP_TOLERANCE): float epsilon * 100
    assert val == pytest.approx(case["expected"], rel=tol)
    if case["expected_pvalue"] is None:
        assert math.isnan(pv)  # the weighted forms report no significance
    else:
        assert pv == pytest.approx(case["expected_pvalue"], rel=tol)


def test_correlation_vectors_through_the_stripe_comparison():
    """the same six pairs as the stripes of two band matrices: evaluate.compare is the code path
    `python -m modle_amd evaluate` runs"""
    for case in VEC["correlation"]:
        if case["weights"] is not None:
            continue
        n = len(case["v1"])
        # one bin whose vertical stripe holds the vector: column n - 1 of an n x n band matrix
        a = np.zeros(n * n + 1, dtype=np.uint32)
        b = np.zeros(n * n + 1, dtype=np.uint32)
        a[(n - 1) * n:(n - 1) * n + n] = case["v1"]
        b[(n - 1) * n:(n - 1) * n + n] = case["v2"]
        val, pv = evaluate.compare(a, b, n, n, case["method"], "vertical")
        assert val[n - 1] == pytest.approx(case["expected"], rel=case["rel_tolerance"])
        assert pv[n - 1] == pytest.approx(case["expected_pvalue"], rel=case["rel_tolerance"])


def test_weighted_spearman_reduces_to_the_plain_one_with_unit_weights():
    rng = np.random.default_rng(5)
    for _ in range(20):
        a = rng.integers(0, 12, size=30)
        b = rng.integers(0, 12, size=30)
        plain, _ = evaluate.spearman(a, b)
        weighted, pv = evaluate.spearman(a, b, np.ones(30))
        assert weighted == pytest.approx(plain, abs=1e-12) and math.isnan(pv)
        # weight 0 removes an element -- the same as dropping it -- as long as it is not tied with
        # another one (a tied group shares the MEAN weight of its members in the reference's rank
        # formula, correlation_impl.hpp:296-309, so a masked member lowers the ranks of the group)
        ua, ub = rng.permutation(30), rng.permutation(30)
        w = np.ones(30)
        w[[3, 17]] = 0.0
        keep = w > 0
        dropped, _ = evaluate.spearman(ua[keep], ub[keep])
        masked, _ = evaluate.spearman(ua, ub, w)
        assert masked == pytest.approx(dropped, abs=1e-12)


# ---------------------------------------------------------------------------------------------
def _import(intervals_bed=None, barriers_bed="", chrom_sizes='chr0\t100\nchr1\t1000\nchr2\t100\n"chr1\t500\n'):
    cfg = api.make_config()
    return genome.import_genome_text(cfg, chrom_sizes, barriers_bed, intervals_bed)


def test_bed_strip_quotes_valid():
    case = [c for c in VEC["bed_parser"] if c["name"].startswith("BED: strip quotes")][0]
    rec = case["record"]
    # the BED9 line as a barrier record: quoted name / strand / rgb are accepted
    chroms, intervals, stats = _import(barriers_bed=rec["line"] + "\n")
    iv = [i for i in intervals if i["name"] == rec["chrom"]][0]
    assert stats["barriers_imported"] == 1
    assert iv["bar_pos"].tolist() == [(rec["chrom_start"] + rec["chrom_end"] + 1) // 2]
    assert iv["bar_dir"].tolist() == [params.DIR_REV if rec["strand"] == "+" else params.DIR_FWD]
    # an unbalanced quote belongs to the name
    uq = case["unbalanced_quote"]
    chroms, intervals, _ = _import(intervals_bed=uq["line"] + "\n")
    assert [(i["name"], i["start"], i["end"]) for i in intervals] == [(uq["chrom"], 0, 1)]


def test_bed_strip_quotes_invalid():
    case = [c for c in VEC["bed_parser"] if c["name"].startswith("BED: strip quotes")][0]
    for line in case["must_throw"]:
        nf = len(line.split("\t"))
        with pytest.raises(genome.GenomeError):
            if nf <= 3:
                # a quoted number: the BED3 dialect of the genomic-intervals file parses it
                _import(intervals_bed=line + "\n")
            else:
                # a quoted score: parsed by the BED6 dialect of the barrier file (the BED3 dialect
                # stops after chromEnd like the reference's, bed.cpp:132-140), so the record gets
                # its sixth field
                _import(barriers_bed=line + "\t+" * (6 - nf) + "\n")
    # the same record with a plain score is fine: it is the quotes that are rejected
    _, _, stats = _import(barriers_bed="chr1\t0\t1\t.\t0.0\t+\n")
    assert stats["barriers_imported"] == 1


def test_bed_parser_crlf():
    case = [c for c in VEC["bed_parser"] if c["name"] == "BED Parser CRLF"][0]
    _, with_crlf, _ = _import(intervals_bed="".join(case["file_lines"]))
    _, plain, _ = _import(intervals_bed="".join(r + "\n" for r in case["expected_records"]))
    key = lambda ivs: [(i["name"], i["start"], i["end"]) for i in ivs]
    assert key(with_crlf) == key(plain) == [(r.split("\t")[0], 0, 1) for r in case["expected_records"]]
    # ... and as a chrom.sizes / barrier file
    _, a, sa = _import(barriers_bed="chr1\t0\t10\tn\t0.5\t+\r\nchr2\t4\t8\tm\t0.25\t-\r\n",
                       chrom_sizes="chr1\t1000\r\nchr2\t100\r\n")
    _, b, sb = _import(barriers_bed="chr1\t0\t10\tn\t0.5\t+\nchr2\t4\t8\tm\t0.25\t-\n",
                       chrom_sizes="chr1\t1000\nchr2\t100\n")
    assert sa == sb and [i["bar_pos"].tolist() for i in a] == [i["bar_pos"].tolist() for i in b] == [[5], [6]]


# ---------------------------------------------------------------------------------------------
def test_bigwig_writer_vector(tmp_path):
    case = VEC["bigwig_writer"][0]
    chroms = list(zip(case["chrom_names"], case["chrom_sizes"]))
    bs = case["bin_size"]
    path = str(tmp_path / "test.bw")
    with bigwig.BigWigWriter(path, chroms) as w:
        for name, size in chroms:
            w.write_range(name, np.arange(size // bs, dtype=np.float32), bs, bs)
    bw = BigWig(path)
    assert bw.chroms == chroms
    for name, size in chroms:
        got = bw.query(name, 0, size)
        assert len(got) == (size + bs - 1) // bs
        for i, (start, end, value) in enumerate(got):
            assert (start, end, value) == (bs * i, bs * (i + 1), float(i))

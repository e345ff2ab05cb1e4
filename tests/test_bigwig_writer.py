"""bigWig writer for the 1-D LEF occupancy track (reference:
simulation.cpp:170-197): the file parses by the published layout (tests/bigwig_reader.py walks
header, B+ tree, R-tree and zlib sections on its own) and holds occupancy / max(occupancy) as
float32 with span = step = bin size, chromosome by chromosome in genome order."""
import numpy as np
import pytest

from bigwig_reader import BigWig
from modle_amd import bigwig


def test_occupancy_track_round_trip(tmp_path):
    rng = np.random.default_rng(4)
    chroms = [("chr1", 1_003_000), ("chrNoData", 5_000), ("chr2", 70_000_000), ("chrLongName_random", 12_345)]
    bin_size = 5000
    path = str(tmp_path / "occ.bw")
    occ1 = rng.integers(0, 1000, size=-(-chroms[0][1] // bin_size)).astype(np.uint64)
    occ2 = rng.integers(0, 50, size=-(-chroms[2][1] // bin_size)).astype(np.uint64)  # 14000 bins: 2 sections
    sub = rng.integers(1, 9, size=3).astype(np.uint64)
    with bigwig.BigWigWriter(path, chroms) as w:
        w.write_occupancy("chr1", occ1, bin_size)
        w.write_occupancy("chr2", occ2, bin_size)
        w.write_occupancy("chrLongName_random", sub, bin_size, offset_bp=0)
        with pytest.raises(bigwig.BigWigError):  # out of genome order
            w.write_occupancy("chr1", occ1, bin_size)
    bw = BigWig(path)
    assert bw.version == 4 and bw.zoom_levels == 0
    assert bw.chroms == chroms
    secs = bw.sections()
    assert bw.n_sections == len(secs) == 1 + 2 + 1
    assert all(step == span == bin_size for _, _, _, step, span, _ in secs)
    got1 = np.array(secs[0][5], dtype=np.float32)
    exp1 = (occ1.astype(np.float64) / float(occ1.max())).astype(np.float32)
    assert secs[0][0] == 0 and secs[0][1] == 0 and np.array_equal(got1, exp1)
    got2 = np.array(secs[1][5] + secs[2][5], dtype=np.float32)
    assert np.array_equal(got2, (occ2.astype(np.float64) / float(occ2.max())).astype(np.float32))
    assert secs[1][0] == secs[2][0] == 2 and secs[2][1] == 8186 * bin_size
    # the last bin of a chromosome is clipped to its end
    assert secs[0][2] == chroms[0][1] and secs[3][2] == chroms[3][1]
    # a range query through the R-tree
    q = bw.query("chr2", 41_000_000, 41_012_000)
    assert [(a, b) for a, b, _ in q] == [(41_000_000 + 5000 * i, 41_005_000 + 5000 * i) for i in range(3)]
    assert [v for _, _, v in q] == [float(x) for x in got2[8200:8203]]
    assert bw.query("chrNoData", 0, 5000) == []
    # total summary: bases covered and extrema
    assert bw.summary["bases"] == chroms[0][1] + chroms[2][1] + chroms[3][1]
    assert bw.summary["max"] == 1.0 and bw.summary["min"] == float(min(got1.min(), got2.min()))


def test_many_sections_make_a_two_level_index(tmp_path):
    """more than 256 sections: the R-tree gets an inner level"""
    chroms = [("c", 300 * 8186 * 10)]
    vals = np.arange(300 * 8186, dtype=np.float32) % 977
    path = str(tmp_path / "big.bw")
    with bigwig.BigWigWriter(path, chroms) as w:
        w.write_range("c", vals, 10, 10)
    bw = BigWig(path)
    assert bw.n_sections == 300
    q = bw.query("c", 2_999_999 * 8, 2_999_999 * 8 + 25)
    first = (2_999_999 * 8) // 10
    assert [v for _, _, v in q] == [float(vals[first + i]) for i in range(len(q))] and len(q) in (3, 4)

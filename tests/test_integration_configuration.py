"""The reference's own integration configuration as a parity case, end to end.

Shape of test/integration/src/modle_integration_suite/cli/modle.py:40-75:

    modle simulate --chrom-sizes ... --genomic-intervals regions.bed --extrusion-barrier-file ...
        --output-prefix ... --resolution 20kb --verbose --target-contact-density 20 --ncells 2
        --track-1d-lef-position --max-burnin-epochs 5000 --threads N

with SEVERAL windows per chromosome in the regions BED, compared the way the reference's validators
do (validators/cooler.py:29-84: same chromosomes, same bins, same pixel coordinates per chromosome,
counts isclose(rtol=1e-5); validators/bigwig.py:15-52: same chromosomes, same number of intervals
per chromosome, values isclose(rtol=1e-5)).

The reference's golden files (modle_sim_reference_001.cool / .bw over grch38 inputs) live in its
Zenodo data set, which is not in the image: the expected tables here come from the CPU oracle on a
synthetic genome of the same shape, built by code that shares nothing with the writers under
test.  Point MODLE_GOLDEN_PREFIX / MODLE_GOLDEN_DATA_DIR at the real files and the last test
compares against them with the same rule."""
import os

import numpy as np
import pytest

from bigwig_reader import BigWig
from modle_amd import api, cli, driver, genome
from test_cooler_writer import _h5py_read

SIZES = [("chrA", 12_000_000), ("chrB", 3_000_000), ("chrC", 8_000_000), ("chrD", 2_000_000)]
# several windows on chrA, none on chrB (it must still be in both files), one on chrC, and one on
# chrD, which has no barriers and is therefore not simulated (scheduler_simulate.cpp:111-124)
WINDOWS = [("chrA", 1_000_000, 4_000_000), ("chrA", 6_000_000, 9_500_000), ("chrA", 11_000_000, 12_000_000),
           ("chrC", 500_000, 3_500_000), ("chrD", 0, 2_000_000)]


def reference_args(tmp_path, prefix):
    return ["simulate", "--chrom-sizes", str(tmp_path / "g.chrom.sizes"), "--genomic-intervals",
            str(tmp_path / "regions.bed"), "--extrusion-barrier-file", str(tmp_path / "barriers.bed"),
            "--output-prefix", prefix, "--resolution", "20kb", "--verbose", "--target-contact-density",
            "20", "--ncells", "2", "--track-1d-lef-position", "--max-burnin-epochs", "5000",
            "--threads", "4"]


def write_inputs(tmp_path):
    rng = np.random.default_rng(17)
    (tmp_path / "g.chrom.sizes").write_text("".join(f"{n}\t{s}\n" for n, s in SIZES))
    (tmp_path / "regions.bed").write_text("".join(f"{c}\t{a}\t{b}\n" for c, a, b in WINDOWS))
    lines = []
    for name, size in SIZES:
        if name == "chrD":
            continue
        for p in sorted(rng.choice(size - 100, size=size // 80_000, replace=False)):
            lines.append(f"{name}\t{p}\t{p + 19}\t.\t{rng.uniform(0.6, 1.0):.3f}\t{'+' if rng.random() < 0.5 else '-'}")
    rng.shuffle(lines)
    (tmp_path / "barriers.bed").write_text("\n".join(lines) + "\n")


def plan_of(tmp_path, argv):
    cfg = cli.config_from_args(cli.build_parser().parse_args(argv))
    chroms, ivs, _ = genome.import_genome(cfg, str(tmp_path / "g.chrom.sizes"), str(tmp_path / "barriers.bed"),
                                          str(tmp_path / "regions.bed"))
    return cfg, chroms, driver.plan_genome(cfg, ivs)


def oracle_outputs(oracle, cfg, plan):
    mats, occs = [], []
    for entry in plan:
        if entry["skipped"]:
            mats.append(None)
            occs.append(None)
            continue
        iv = entry["interval"]
        c, _, o, _ = oracle.simulate_interval(cfg, iv["start"], iv["end"], iv["bar_pos"], iv["bar_dir"],
                                              iv["bar_stp_active"], iv["bar_stp_inactive"], entry["tasks"],
                                              nthreads=2, track_occupancy=True)
        mats.append(c)
        occs.append(o)
    return mats, occs


def expected_tables(cfg, chroms, plan, mats, occs):
    """what the reference writes (contact_matrix_dense_io_impl.hpp:51-150, simulation.cpp:170-197),
    restated from the matrices: chromosome table, bin table, pixels per chromosome in (bin1, bin2)
    order with genome-wide bin ids, occupancy intervals per chromosome"""
    bs = int(cfg.bin_size)
    bins, first_bin = [], {}
    for k, (name, size) in enumerate(chroms):
        first_bin[name] = len(bins)
        bins += [[k, a, min(a + bs, size)] for a in range(0, size, bs)]
    pixels = {name: [] for name, _ in chroms}
    tracks = {name: [] for name, _ in chroms}
    for entry, m, o in zip(plan, mats, occs):
        if m is None:
            continue
        iv = entry["interval"]
        nrows, ncols = entry["nrows"], entry["ncols"]
        off = first_bin[iv["name"]] + iv["start"] // bs
        for i in range(ncols):
            for j in range(i, min(ncols, i + nrows)):
                n = int(m[j * nrows + (j - i)])
                if n != 0:
                    pixels[iv["name"]].append([off + i, off + j, n])
        mx = float(o.max())
        size = dict(chroms)[iv["name"]]
        for i in range(ncols):
            a = iv["start"] + i * bs
            tracks[iv["name"]].append((a, min(a + bs, size), np.float32(float(o[i]) / mx)))
    return {"chroms": [list(c) for c in chroms], "bins": bins, "pixels_by_chrom": pixels, "tracks": tracks}


def compare_coolers_like_the_reference(expected, found, rtol=1.0e-5):
    """validators/cooler.py:29-84 on plain tables; returns the dict of errors (empty = pass)"""
    if expected["chroms"] != found["chroms"]:
        return {"chromosomes differ": f"expected {expected['chroms']}, found {found['chroms']}"}
    if expected["bins"] != found["bins"]:
        return {"found differences in bin coordinates": ""}
    errors = {}
    for chrom, _ in expected["chroms"]:
        e, f = expected["pixels_by_chrom"][chrom], found["pixels_by_chrom"][chrom]
        if len(e) != len(f):
            errors[f"{chrom}: pixel table has an unexpected number of records"] = f"expected {len(e)}, found {len(f)}"
            continue
        if [r[:2] for r in e] != [r[:2] for r in f]:
            errors[f"{chrom}: found differences in pixel coordinates"] = ""
            continue
        ce, cf = np.array([r[2] for r in e], dtype=float), np.array([r[2] for r in f], dtype=float)
        bad = int((~np.isclose(ce, cf, rtol=rtol)).sum())
        if bad:
            errors[f"{chrom}: found differences in pixel counts"] = f"found {bad} differences"
    return errors


def compare_bigwigs_like_the_reference(expected_tracks, chroms, bw, rtol=1.0e-5):
    """validators/bigwig.py:15-52"""
    if bw.chroms != chroms:
        return {"chromosomes differ": f"{bw.chroms}"}
    errors = {}
    for name, size in chroms:
        got, exp = bw.query(name, 0, size), expected_tracks[name]
        if len(got) != len(exp):
            errors[f"{name}: unexpected number of entries"] = f"expected {len(exp)}, found {len(got)}"
            continue
        if [(a, b) for a, b, _ in got] != [(a, b) for a, b, _ in exp]:
            errors[f"{name}: interval coordinates differ"] = ""
            continue
        if len(exp) and (~np.isclose([v for _, _, v in exp], [v for _, _, v in got], rtol=rtol)).any():
            errors[f"{name}: found differences in values"] = ""
    return errors


def check_outputs(cfg, chroms, plan, mats, occs, prefix):
    exp = expected_tables(cfg, chroms, plan, mats, occs)
    got = _h5py_read(prefix + ".cool")
    assert compare_coolers_like_the_reference(exp, got) == {}
    # beyond the validator's tolerance: the counts are integers and must be identical
    assert got["pixels_by_chrom"] == exp["pixels_by_chrom"]
    assert compare_bigwigs_like_the_reference(exp["tracks"], chroms, BigWig(prefix + "_lef_1d_occupancy.bw")) == {}
    # the shape of the run: three windows of chrA in ONE chromosome's pixel range, chrB and chrD
    # present with all their bins and no pixel
    assert [c[0] for c in got["chroms"]] == [n for n, _ in SIZES]
    assert len(got["bins"]) == sum(-(-s // 20_000) for _, s in SIZES)
    assert got["pixels_by_chrom"]["chrB"] == [] and got["pixels_by_chrom"]["chrD"] == []
    a = np.array(got["pixels_by_chrom"]["chrA"])
    for _, lo, hi in WINDOWS[:3]:
        sel = (a[:, 0] >= lo // 20_000) & (a[:, 0] < hi // 20_000)
        assert sel.any() and (a[sel, 1] < -(-hi // 20_000)).all()  # no pixel leaves its window
    return exp


def test_reference_integration_configuration_with_the_oracle(oracle, tmp_path):
    """host logic (options with units, regions BED, plan), oracle, the two writers, the two
    readers: everything but the kernel, in the reference's integration shape"""
    write_inputs(tmp_path)
    prefix = str(tmp_path / "001" / "modle_sim_001")
    argv = reference_args(tmp_path, prefix)
    cfg, chroms, plan = plan_of(tmp_path, argv)
    assert (cfg.bin_size, cfg.num_cells, cfg.target_contact_density, cfg.max_burnin_epochs,
            cfg.track_1d_lef_position) == (20_000, 2, 20.0, 5000, 1)
    assert [(e["interval"]["name"], e["interval"]["start"], e["interval"]["end"], e["skipped"]) for e in plan] == \
        [(c, a, b, c == "chrD") for c, a, b in WINDOWS]
    mats, occs = oracle_outputs(oracle, cfg, plan)
    os.makedirs(os.path.dirname(prefix))
    driver.write_cooler(prefix + ".cool", cfg, plan, mats, chroms=chroms)
    driver.write_bigwig(prefix + "_lef_1d_occupancy.bw", cfg, plan, occs, chroms)
    check_outputs(cfg, chroms, plan, mats, occs, prefix)
    # without the genome's chromosome list only the chromosomes of the plan are known: the file
    # then lacks chrB (what the front end did before it passed `chroms`)
    driver.write_cooler(prefix + "_plan_only.cool", cfg, plan, mats)
    assert [c[0] for c in _h5py_read(prefix + "_plan_only.cool")["chroms"]] == ["chrA", "chrC", "chrD"]


@pytest.mark.gpu
def test_reference_integration_configuration_on_the_gpu(oracle, tmp_path):
    """the same command line through `python -m modle_amd simulate` on the MI355X: .cool and .bw
    equal the oracle's by the reference validators' rules (and exactly)"""
    write_inputs(tmp_path)
    prefix = str(tmp_path / "001" / "modle_sim_001")
    argv = reference_args(tmp_path, prefix)
    assert cli.main(argv) == 0
    cfg, chroms, plan = plan_of(tmp_path, argv)
    mats, occs = oracle_outputs(oracle, cfg, plan)
    check_outputs(cfg, chroms, plan, mats, occs, prefix)


@pytest.mark.gpu
@pytest.mark.skipif(not os.environ.get("MODLE_GOLDEN_PREFIX"),
                    reason="the reference's golden files (Zenodo test data set) are not in the image: set "
                           "MODLE_GOLDEN_DATA_DIR (grch38.chrom.sizes, grch38_regions_of_interest.bed, "
                           "grch38_h1_extrusion_barriers.bed.xz) and MODLE_GOLDEN_PREFIX (modle_sim_reference_001)")
def test_against_the_reference_golden_files(tmp_path):
    data, golden = os.environ["MODLE_GOLDEN_DATA_DIR"], os.environ["MODLE_GOLDEN_PREFIX"]
    prefix = str(tmp_path / "001" / "modle_sim_001")
    argv = ["simulate", "--chrom-sizes", os.path.join(data, "grch38.chrom.sizes"), "--genomic-intervals",
            os.path.join(data, "grch38_regions_of_interest.bed"), "--extrusion-barrier-file",
            os.path.join(data, "grch38_h1_extrusion_barriers.bed.xz"), "--output-prefix", prefix,
            "--resolution", "20kb", "--target-contact-density", "20", "--ncells", "2",
            "--track-1d-lef-position", "--max-burnin-epochs", "5000", "-q"]
    assert cli.main(argv) == 0
    exp, got = _h5py_read(golden + ".cool"), _h5py_read(prefix + ".cool")
    assert compare_coolers_like_the_reference(exp, got) == {}
    gbw, bw = BigWig(golden + "_lef_1d_occupancy.bw"), BigWig(prefix + "_lef_1d_occupancy.bw")
    tracks = {n: gbw.query(n, 0, s) for n, s in gbw.chroms}
    assert compare_bigwigs_like_the_reference(tracks, gbw.chroms, bw) == {}

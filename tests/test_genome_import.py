"""Genome import and the `modle simulate`-shaped front end.

Fixtures: (chrom.sizes, BED6 [, BED3]) texts -> (chromosomes, intervals, barrier arrays, task
list), expected values worked out by hand from the reference's rules (genome.cpp:260-271,
423-469 barrier position / strand / score; bed.cpp:245-320, 428-586 record parsing, header
skipping, duplicate detection; chrom_sizes.cpp; scheduler_simulate.cpp:104-160 task list)."""
import gzip
import lzma
import os

import numpy as np
import pytest

from modle_amd import api, cli, driver, genome
from modle_amd.params import DIR_FWD, DIR_REV

CHROM_SIZES = "chr1\t1000000\n'chr2'\t500000\n\nchrEmpty\t20000\n"
BARRIERS = """# a comment line in the header
track name=ctcf description="header line"
chr1\t100\t120\tb0\t0.9\t+
chr1\t5000\t5019\tb1\t0\t-
chr1 7000 7019 b2 0.5 .
chr2\t499990\t500000\tb3\t1\t-

"chr1"\t300\t301\tb4\t0.75\tplus
chrUn\t1\t2\tb5\t0.5\t+
chr2\t10\t20\tb6\t0.25\tREV\textra1\textra2
"""


def cfg():
    return api.make_config(num_cells=3, seed=11)


def test_fixture_whole_chromosomes():
    c = cfg()
    chroms, ivs, stats = genome.import_genome_text(c, CHROM_SIZES, BARRIERS)
    assert chroms == [("chr1", 1_000_000), ("chr2", 500_000), ("chrEmpty", 20_000)]
    assert [(iv["name"], iv["start"], iv["end"]) for iv in ivs] == \
        [("chr1", 0, 1_000_000), ("chr2", 0, 500_000), ("chrEmpty", 0, 20_000)]
    assert stats == {"barriers_imported": 5, "barriers_without_strand": 1}
    # positions: (start + end + 1) / 2, in file order
    assert ivs[0]["bar_pos"].tolist() == [(100 + 120 + 1) // 2, (5000 + 5019 + 1) // 2, (300 + 301 + 1) // 2]
    assert ivs[0]["bar_pos"].tolist() == [110, 5010, 301]
    # strand '+' / plus => blocks rev-moving units; '-' / REV => fwd (extrusion_barriers_impl.hpp:61-72)
    assert ivs[0]["bar_dir"].tolist() == [DIR_REV, DIR_FWD, DIR_REV]
    puu, pbb = c.barrier_not_occupied_stp, c.barrier_occupied_stp
    exp = [api.stp_active_from_occupancy(puu, 0.9), pbb, api.stp_active_from_occupancy(puu, 0.75)]
    assert ivs[0]["bar_stp_active"].tolist() == exp  # score 0 => the default
    assert ivs[0]["bar_stp_inactive"].tolist() == [puu] * 3
    assert ivs[1]["bar_pos"].tolist() == [(499990 + 500000 + 1) // 2, 15]
    assert ivs[1]["bar_dir"].tolist() == [DIR_FWD, DIR_FWD]
    assert ivs[1]["bar_stp_active"].tolist() == [api.stp_active_from_occupancy(puu, 1.0),
                                                 api.stp_active_from_occupancy(puu, 0.25)]
    assert len(ivs[2]["bar_pos"]) == 0
    # task list: the chromosome without barriers is skipped, ids keep counting per simulated interval
    plan = driver.plan_genome(c, ivs)
    assert [e["skipped"] for e in plan] == [False, False, True]
    t1, t2 = plan[0]["tasks"], plan[1]["tasks"]
    assert [t.id for t in t1] + [t.id for t in t2] == [0, 1, 2, 3, 4, 5]
    assert [t.cell_id for t in t2] == [0, 1, 2]
    assert t1[0].num_lefs == api.compute_num_lefs(c, 1_000_000) == 20
    st = api.prng_seed(api.interval_hash("chr2", 500_000, 0, 500_000, 11))
    assert list(t2[0].prng) == st and list(t2[1].prng) == api.prng_jump(st)
    nr, nc = api.matrix_shape(c, 500_000)
    tot = round(nr * nc * c.target_contact_density)
    assert sum(t.num_target_contacts for t in t2) == tot


def test_fixture_genomic_intervals():
    c = cfg()
    intervals = "chr1\t0\t200\nchr2\t499000\t500000\nchr1\t4000\t6000\n"
    chroms, ivs, stats = genome.import_genome_text(c, CHROM_SIZES, BARRIERS, intervals)
    # genome order: chromosome order of chrom.sizes, then by start
    assert [(iv["name"], iv["start"], iv["end"]) for iv in ivs] == \
        [("chr1", 0, 200), ("chr1", 4000, 6000), ("chr2", 499000, 500000)]
    assert [iv["bar_pos"].tolist() for iv in ivs] == [[110], [5010], [499995]]
    assert stats["barriers_imported"] == 3
    plan = driver.plan_genome(c, ivs)
    assert plan[1]["tasks"][0].num_lefs == api.compute_num_lefs(c, 2000)
    st = api.prng_seed(api.interval_hash("chr1", 1_000_000, 4000, 6000, 11))
    assert list(plan[1]["tasks"][0].prng) == st


@pytest.mark.parametrize("text,needle", [
    ("chr1\t10\t20\tx\t0.5\t+\nchr1\t10\t20\ty\t0.7\t-\n", "duplicate"),
    ("chr1\t10\t20\tx\t1.5\t+\n", "score between 0 and 1"),
    ("chr1\t10\t20\tx\t0.5\t?\n", "unrecognized strand"),
    ("chr1\t30\t20\tx\t0.5\t+\n", "chrom_start > chrom_end"),
    ("chr1\t10\t20\tx\t0.5\n", "at least 6 fields"),
    ("chr1\tten\t20\tx\t0.5\t+\n", "convert"),
    ("chr1\t10\t20\tx\t2000\t+\n", "between 0.0 and 1000.0"),
])
def test_malformed_barrier_records(text, needle):
    with pytest.raises(genome.GenomeError) as e:
        genome.import_genome_text(cfg(), CHROM_SIZES, text)
    assert needle in str(e.value)


@pytest.mark.parametrize("text,needle", [
    ("chr1\t100\nchr1\t200\n", "multiple records"),
    ("chr1\t0\n", "length of 0bp"),
    ("chr1 100\n", "exactly 2 fields"),
    ("", "Unable to import any chromosome"),
])
def test_malformed_chrom_sizes(text, needle):
    with pytest.raises(genome.GenomeError) as e:
        genome.import_genome_text(cfg(), text, BARRIERS)
    assert needle in str(e.value)


def test_name_as_not_bound_stp_is_validated():
    ok = "chr1\t10\t20\t0.4\t0.5\t+\n"
    genome.import_genome_text(cfg(), CHROM_SIZES, ok, interpret_name_as_not_bound_stp=True)
    with pytest.raises(genome.GenomeError) as e:
        genome.import_genome_text(cfg(), CHROM_SIZES, BARRIERS, interpret_name_as_not_bound_stp=True)
    assert "invalid name field" in str(e.value)


def test_compressed_inputs(tmp_path):
    p1, p2 = tmp_path / "g.chrom.sizes", tmp_path / "b.bed.xz"
    p1.write_bytes(gzip.compress(CHROM_SIZES.encode()))
    p2.write_bytes(lzma.compress(BARRIERS.encode()))
    chroms, ivs, _ = genome.import_genome(cfg(), str(p1), str(p2))
    assert len(chroms) == 3 and ivs[0]["bar_pos"].tolist() == [110, 5010, 301]


def test_cli_config_matches_the_reference_derivations():
    ap = cli.build_parser()
    a = ap.parse_args(["simulate", "-c", "x", "-b", "y", "-o", "z"])
    c = cli.config_from_args(a)
    d = api.make_config()
    assert bytes(c) == bytes(d)  # no options: the reference defaults
    a = ap.parse_args(["sim", "-c", "x", "-b", "y", "-o", "z", "-r", "10000", "--ncells", "7",
                       "--lef-density", "64", "--fwd-extrusion-speed", "3000",
                       "--contact-sampling-strategy", "loop-only", "--extrusion-barrier-occupancy",
                       "0.9", "--no-track-1d-lef-position", "--seed", "5", "--skip-burnin"])
    c = cli.config_from_args(a)
    assert c.bin_size == 10000 and c.num_cells == 7 and c.number_of_lefs_per_mbp == 64.0
    assert c.fwd_extrusion_speed == 3000 and c.rev_extrusion_speed == 8000  # 0.8 * resolution
    assert c.contact_sampling_strategy == 4 and c.tad_to_loop_contact_ratio == 0.0
    assert c.track_1d_lef_position == 0 and c.seed == 5 and c.skip_burnin == 1
    assert c.barrier_occupied_stp == api.stp_active_from_occupancy(c.barrier_not_occupied_stp, 0.9)
    a = ap.parse_args(["sim", "-c", "x", "-b", "y", "-o", "z", "-s", "simulation-epochs",
                       "--target-number-of-epochs", "50"])
    c = cli.config_from_args(a)
    assert c.target_contact_density < 0 and c.target_simulation_epochs == 50
    with pytest.raises(SystemExit):
        cli.config_from_args(ap.parse_args(["sim", "-c", "x", "-b", "y", "-o", "z",
                                            "--target-number-of-epochs", "50"]))
    with pytest.raises(SystemExit):
        cli.config_from_args(ap.parse_args(["sim", "-c", "x", "-b", "y", "-o", "z",
                                            "--extrusion-barrier-occupancy", "0.9",
                                            "--extrusion-barrier-bound-stp", "0.9"]))
    assert cli.output_paths("out/p") == ("out/p.cool", "out/p_lef_1d_occupancy.bw")


@pytest.mark.gpu
def test_cli_simulate_end_to_end(oracle, tmp_path):
    """files in -> .cool and .bw out through the front end; the cooler's pixels equal the
    oracle's matrices for the same genome and options, the bigWig holds occupancy / max"""
    from bigwig_reader import BigWig
    from test_cooler_writer import _h5py_read

    rng = np.random.default_rng(8)
    sizes = "chrA\t3000000\nchrB\t1000000\nchrC\t2500000\n"
    lines = []
    for name, size in (("chrA", 3_000_000), ("chrC", 2_500_000)):
        for p in sorted(rng.choice(size - 100, size=30, replace=False)):
            lines.append(f"{name}\t{p}\t{p + 19}\t.\t{rng.uniform(0.6, 1.0):.3f}\t{'+' if rng.random() < 0.5 else '-'}")
    rng.shuffle(lines)  # BED order is not position order: the library sorts per interval
    (tmp_path / "g.chrom.sizes").write_text(sizes)
    (tmp_path / "b.bed").write_text("\n".join(lines) + "\n")
    prefix = str(tmp_path / "out" / "run")
    argv = ["simulate", "-c", str(tmp_path / "g.chrom.sizes"), "-b", str(tmp_path / "b.bed"), "-o", prefix,
            "--ncells", "4", "-w", "1000000", "--target-contact-density", "0.2", "--seed", "3", "-q"]
    assert cli.main(argv) == 0
    with pytest.raises(SystemExit):  # outputs exist
        cli.main(argv)
    assert cli.main(argv + ["--force"]) == 0
    cfg = cli.config_from_args(cli.build_parser().parse_args(argv))
    chroms, ivs, _ = genome.import_genome(cfg, str(tmp_path / "g.chrom.sizes"), str(tmp_path / "b.bed"))
    plan = driver.plan_genome(cfg, ivs)
    got = _h5py_read(prefix + ".cool")
    bw = BigWig(prefix + "_lef_1d_occupancy.bw")
    assert got["chroms"] == [list(c) for c in chroms] and bw.chroms == chroms
    first_bin = 0
    for entry in plan:
        iv = entry["interval"]
        nbins = -(-iv["size"] // int(cfg.bin_size))
        if entry["skipped"]:
            assert got["pixels_by_chrom"][iv["name"]] == []
            first_bin += nbins
            continue
        oc, om, oo, _ = oracle.simulate_interval(cfg, iv["start"], iv["end"], iv["bar_pos"], iv["bar_dir"],
                                                 iv["bar_stp_active"], iv["bar_stp_inactive"],
                                                 entry["tasks"], nthreads=4)
        nrows, ncols = entry["nrows"], entry["ncols"]
        dense = np.zeros(nrows * ncols, dtype=np.int64)
        for b1, b2, n in got["pixels_by_chrom"][iv["name"]]:
            i, j = b1 - first_bin, b2 - first_bin
            dense[j * nrows + (j - i)] = n
        assert np.array_equal(dense, oc[:nrows * ncols].astype(np.int64)), iv["name"]
        vals = np.array([v for _, _, v in bw.query(iv["name"], 0, iv["size"])], dtype=np.float32)
        assert np.array_equal(vals, (oo.astype(np.float64) / float(oo.max())).astype(np.float32))
        first_bin += nbins
    assert got["attrs"]["assembly"] == "unknown" and "modle_amd" in got["attrs"]["generated-by"]


def test_records_outside_the_simulated_intervals_are_not_validated():
    """the reference validates score / name only for the records its interval tree returns for a
    simulated interval (genome.cpp:470-489): a bad score on a stretch that --genomic-intervals
    leaves out does not stop the import"""
    sizes = "chr1\t100000\nchr2\t50000\n"
    bad_outside = "chr1\t100\t120\tx\t0.5\t+\nchr1\t60000\t60020\ty\t7.5\t-\nchr2\t10\t20\tz\t3\t+\n"
    chroms, ivs, stats = genome.import_genome_text(cfg(), sizes, bad_outside, "chr1\t0\t50000\n")
    assert [(i["name"], i["start"], i["end"], i["bar_pos"].tolist()) for i in ivs] == [("chr1", 0, 50000, [110])]
    assert stats["barriers_imported"] == 1
    # the same file with the whole genome simulated: both bad records are looked at
    with pytest.raises(genome.GenomeError) as e:
        genome.import_genome_text(cfg(), sizes, bad_outside)
    assert "invalid score field" in str(e.value)
    # a record that overlaps the window is validated even though its midpoint lies outside
    with pytest.raises(genome.GenomeError):
        genome.import_genome_text(cfg(), sizes, "chr1\t49990\t50030\tq\t9\t+\n", "chr1\t0\t50000\n")
    _, ivs, stats = genome.import_genome_text(cfg(), sizes, "chr1\t49990\t50030\tq\t0.9\t+\n", "chr1\t0\t50000\n")
    assert ivs[0]["bar_pos"].tolist() == [] and stats["barriers_imported"] == 0


def test_record_straddling_an_interval_edge_is_left_out():
    """Known divergence from the reference's RELEASE builds: a record that
    overlaps an interval while its midpoint lies outside is kept there (add_extrusion_barriers only
    asserts, genome.cpp:288-297, 470-489) and draws a barrier state per epoch; the importer leaves it
    out, because the device layout takes barrier positions inside the interval only."""
    sizes = "chr1\t100000\n"
    windows = "chr1\t0\t50000\nchr1\t50000\t100000\n"
    # midpoint (49990 + 50030 + 1) / 2 = 50010: inside the second window, outside the first although
    # the record overlaps both; the 1-bp record on the last base has its midpoint AT the end
    bed = "chr1\t49990\t50030\tedge\t0.9\t+\nchr1\t99999\t100000\tlast\t0.8\t-\nchr1\t10\t30\tin\t0.7\t+\n"
    _, ivs, stats = genome.import_genome_text(cfg(), sizes, bed, windows)
    assert [(i["start"], i["end"], i["bar_pos"].tolist()) for i in ivs] == [(0, 50000, [20]), (50000, 100000, [50010])]
    assert stats["barriers_imported"] == 2
    # ... and what comes out is accepted by the device path's own range check as is
    for iv in ivs:
        assert all(iv["start"] <= p < iv["end"] for p in iv["bar_pos"].tolist())

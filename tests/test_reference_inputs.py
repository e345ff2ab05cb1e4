"""The importer and the task generation on the reference's REAL inputs.

`/root/reference/examples/data/hg38.chrom.sizes` and `hg38_extrusion_barriers.bed.xz` (the inputs of
the reference's README run and of BASELINE configs 1-4) exist in the build container only and are
never copied: these tests read them where they lie and skip everywhere else (the GPU box has no
/root/reference).  They pin

* `modle_genome_import` on the real files: 24 chromosomes, 38 815 barriers (19 666 `+` / 19 149 `-`),
  chr1 3 518, occupancies within [0.60, 1.00], sum of `compute_num_lefs` 61 766 -- the figures
  SURVEY.md section 8 quotes (reference: src/libmodle/internal/genome.cpp:423-469, 260-271);
* the per-chromosome table of `modle_amd.synthetic` (what the benchmark's synthetic genome is shaped
  by) against the real file;
* that no record of the real file is touched by the importer's known divergence (INTEGRATION.md
  section 5: records whose midpoint lies outside the interval);
* two chr21 cells through the oracle with the task list of `modle_hip_make_tasks` on the real
  barriers: contact conservation and the per-cell targets of scheduler_simulate.cpp:129-141.
"""
import lzma
import os

import numpy as np
import pytest

from modle_amd import api, driver, genome, synthetic
from modle_amd.params import DIR_FWD, DIR_REV

DATA = "/root/reference/examples/data"
SIZES = os.path.join(DATA, "hg38.chrom.sizes")
BARRIERS = os.path.join(DATA, "hg38_extrusion_barriers.bed.xz")

pytestmark = pytest.mark.skipif(not (os.path.exists(SIZES) and os.path.exists(BARRIERS)),
                                reason="the reference's example data is not here (build container only)")


@pytest.fixture(scope="module")
def real():
    cfg = api.make_config()  # the reference's defaults
    chroms, ivs, stats = genome.import_genome(cfg, SIZES, BARRIERS)
    return cfg, chroms, ivs, stats


def test_importer_on_the_real_files(real):
    cfg, chroms, ivs, stats = real
    assert len(chroms) == 24 and len(ivs) == 24
    assert chroms == synthetic.GRCH38  # names, order and lengths
    assert stats == {"barriers_imported": 38815, "barriers_without_strand": 0}
    n_plus = sum(int((iv["bar_dir"] == DIR_REV).sum()) for iv in ivs)   # BED strand '+' blocks rev units
    n_minus = sum(int((iv["bar_dir"] == DIR_FWD).sum()) for iv in ivs)
    assert (n_plus, n_minus) == (19666, 19149)
    assert len(ivs[0]["bar_pos"]) == 3518 and len(ivs[-1]["bar_pos"]) == 26
    assert sum(api.compute_num_lefs(cfg, iv["end"] - iv["start"]) for iv in ivs) == 61766
    assert api.compute_num_lefs(cfg, ivs[0]["size"]) == 4979
    puu = cfg.barrier_not_occupied_stp
    for iv in ivs:
        assert (iv["start"], iv["end"]) == (0, iv["size"])
        assert np.all(iv["bar_stp_inactive"] == puu)
        occ = np.array([api.lib().modle_hip_occupancy_from_stp(a, puu) for a in iv["bar_stp_active"]])
        assert occ.min() >= 0.60 - 1e-9 and occ.max() <= 1.0 + 1e-9, (iv["name"], occ.min(), occ.max())
        pos = iv["bar_pos"].astype(np.int64)
        assert np.all(pos >= 0) and np.all(pos < iv["size"])


def test_synthetic_genome_has_the_real_per_chromosome_shape(real):
    _, _, ivs, _ = real
    assert {iv["name"]: len(iv["bar_pos"]) for iv in ivs} == synthetic.GRCH38_H1_BARRIERS
    synth = synthetic.grch38_like(seed=42)
    assert [(s["name"], s["size"], len(s["bar_pos"])) for s in synth] == \
        [(iv["name"], iv["size"], len(iv["bar_pos"])) for iv in ivs]
    real_occ = np.concatenate([[api.lib().modle_hip_occupancy_from_stp(a, b) for a, b in
                                zip(iv["bar_stp_active"], iv["bar_stp_inactive"])] for iv in ivs])
    synth_occ = np.concatenate([s["bar_occupancy"] for s in synth])
    assert abs(real_occ.mean() - synth_occ.mean()) < 0.01 and abs(real_occ.std() - synth_occ.std()) < 0.015


def test_no_record_of_the_real_file_meets_the_known_divergence():
    """every record's midpoint lies inside its chromosome"""
    sizes = dict(line.split()[:2] for line in open(SIZES).read().splitlines() if line.strip())
    n = 0
    with lzma.open(BARRIERS, "rt") as fh:
        for line in fh:
            f = line.split()
            if len(f) < 6 or f[0].startswith("#"):
                continue
            mid = (int(f[1]) + int(f[2]) + 1) // 2
            assert 0 <= mid < int(sizes[f[0]]), line
            n += 1
    assert n == 38815


def test_two_chr21_cells_on_the_real_barriers_through_the_oracle(real, oracle):
    cfg0, _, ivs, _ = real
    cfg = api.make_config(num_cells=512)  # the README run's cell count: its target-contact split
    iv = next(i for i in ivs if i["name"] == "chr21")
    plan = driver.plan_genome(cfg, [iv])
    tasks = api.slice_tasks(plan[0]["tasks"], 0, 2)
    assert tasks[0].num_lefs == 934 and len(iv["bar_pos"]) == 427
    c, missed, occ, res = oracle.simulate_interval(cfg, iv["start"], iv["end"], iv["bar_pos"], iv["bar_dir"],
                                                   iv["bar_stp_active"], iv["bar_stp_inactive"], tasks,
                                                   nthreads=2)
    assert [r.num_contacts for r in res] == [t.num_target_contacts for t in tasks]
    assert int(c.astype(np.int64).sum()) + missed == sum(r.num_contacts for r in res)
    assert all(r.burnin_epochs > 100 and r.epochs > r.burnin_epochs for r in res)

"""GPU tests beyond the oracle's reach.

* Sharding on hardware: the cells of a small genome are simulated as one rank and as the two
  shards a 2-GPU run would produce (one after the other on this GPU); the summed shard outputs
  must equal the single-rank outputs word for word.  Together with tests/test_sharding_gloo.py
  (the reduction itself, on CPU) this covers the N > 1 path without a second GPU.
* Full-size properties (BASELINE config 1 scale: a chr1-shaped interval, 4979 LEFs, 3518
  barriers): the oracle would need minutes per cell there, so the checks are size independent:
  the run is reproducible bit for bit, every registered contact is in the matrix or counted as
  missed, the occupancy track holds two entries per sampling event that hit the interval, and
  the cells are statistically sane (burn-in length, contacts per cell = target split)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_plan(cfg, genome, rank, world):
    from modle_amd import api, driver

    plan = driver.plan_genome(cfg, genome, rank, world)
    sim = api.Simulator(cfg, 0)
    try:
        ids = driver.enqueue_plan(sim, cfg, plan)
        sim.launch()
        sim.wait()
        outs = []
        for entry, iid in zip(plan, ids):
            if iid is None:
                outs.append(None)
                continue
            c, missed, occ = sim.copy_outputs(iid)
            res = sim.results(iid) if len(entry["tasks"]) else []
            outs.append((c.astype(np.int64), missed, occ.astype(np.int64),
                         [(r.epochs, r.burnin_epochs, r.num_contacts, r.raws_consumed) for r in res]))
    finally:
        sim.close()
    return outs


def test_two_shards_sum_to_the_single_rank_result():
    from modle_amd import api, synthetic

    genome = [synthetic.synthetic_chromosome("chrA", 6_000_000, seed=1),
              synthetic.synthetic_chromosome("chrB", 2_000_000, seed=2, with_barriers=False),
              synthetic.synthetic_chromosome("chrC", 9_000_000, seed=3)]
    cfg = api.make_config(num_cells=37, seed=7)  # odd count: the shards are ragged (19 + 18)
    whole = _run_plan(cfg, genome, 0, 1)
    s0 = _run_plan(cfg, genome, 0, 2)
    s1 = _run_plan(cfg, genome, 1, 2)
    for w, a, b in zip(whole, s0, s1):
        if w is None:
            assert a is None and b is None
            continue
        assert np.array_equal(w[0], a[0] + b[0])
        assert w[1] == a[1] + b[1]
        assert np.array_equal(w[2], a[2] + b[2])
        assert w[3] == a[3] + b[3]  # per-cell results in cell order: shard 0 then shard 1


@pytest.mark.parametrize("world", [4, 8])
def test_four_and_eight_shards_sum_to_the_single_rank_result(world):
    """SURVEY.md section 8(d)'s gate "identical results for 1/2/4/8-GPU sharding" (the reference's
    independence of the thread count, scheduler_simulate.cpp:129-159): the shards an N-GPU run
    would simulate, one after the other on this GPU, sum to the single-rank outputs word for word,
    and the per-cell results concatenate in cell order."""
    from modle_amd import api, synthetic

    genome = [synthetic.synthetic_chromosome("chrA", 5_000_000, seed=11),
              synthetic.synthetic_chromosome("chrB", 1_500_000, seed=12, with_barriers=False),
              synthetic.synthetic_chromosome("chrC", 7_000_000, seed=13)]
    cfg = api.make_config(num_cells=45, seed=5)  # 45 cells: ragged shards for 4 and for 8 ranks
    whole = _run_plan(cfg, genome, 0, 1)
    shards = [_run_plan(cfg, genome, r, world) for r in range(world)]
    for k, w in enumerate(whole):
        parts = [s[k] for s in shards]
        if w is None:
            assert all(p is None for p in parts)
            continue
        assert np.array_equal(w[0], sum(p[0] for p in parts))
        assert w[1] == sum(p[1] for p in parts)
        assert np.array_equal(w[2], sum(p[2] for p in parts))
        assert w[3] == [r for p in parts for r in p[3]]
        assert all(len(p[3]) in (45 // world, 45 // world + 1) for p in parts)


def test_chr1_scale_properties():
    from modle_amd import api, synthetic

    chrom = synthetic.grch38_like(seed=42, chroms={"chr1"})[0]
    cfg = api.make_config(num_cells=512, seed=0)
    tasks = api.make_tasks(cfg, chrom["name"], chrom["size"], chrom["start"], chrom["end"])
    n = 96
    sub = api.slice_tasks(tasks, 0, n)
    stp_a, stp_i = api.barrier_stps(cfg, chrom["bar_occupancy"])
    runs = []
    for _ in range(2):
        sim = api.Simulator(cfg, 0)
        try:
            runs.append(sim.simulate_interval(chrom["start"], chrom["end"], chrom["bar_pos"],
                                              chrom["bar_dir"], stp_a, stp_i, sub))
        finally:
            sim.close()
    (c0, m0, o0, r0), (c1, m1, o1, r1) = runs
    # reproducible
    assert np.array_equal(c0, c1) and m0 == m1 and np.array_equal(o0, o1)
    assert [(r.epochs, r.raws_consumed) for r in r0] == [(r.epochs, r.raws_consumed) for r in r1]
    # conservation: every registered contact is a matrix increment or a missed update
    assert int(c0.astype(np.int64).sum()) + m0 == sum(r.num_contacts for r in r0)
    # every cell reaches exactly its share of the target contacts (density stopping rule)
    assert [r.num_contacts for r in r0] == [t.num_target_contacts for t in sub]
    # occupancy: two entries per event whose LEF was bound inside the interval
    assert 0 < int(o0.astype(np.int64).sum()) <= 2 * sum(r.sampling_events for r in r0)
    assert int(o0.astype(np.int64).sum()) % 2 == 0
    # burn-in: all LEFs activated (>= target epochs for activation), then a stable window
    for r in r0:
        assert r.burnin_epochs >= cfg.burnin_target_epochs_for_lef_activation
        assert r.epochs > r.burnin_epochs
        assert r.sim_epochs <= r.epochs


def test_chr1_full_launch_properties():
    """BASELINE configs[1] as bench.py launches it: ALL 512 cells of the chr1-shaped interval in one
    launch, which leaves most wave slots empty, so every cell runs on a main wave with a helper and
    a PRNG producer (helper-wave mode: the library's own choice, read back from
    modle_hip_last_launch_info).  The oracle would need minutes for it: checked through the
    size-independent properties of driver.verify_outputs, and against a second launch of the same
    cells with one wave per cell, word for word."""
    import os

    from modle_amd import api, driver, synthetic

    genome = synthetic.grch38_like(seed=42, chroms={"chr1"})
    cfg = api.make_config(num_cells=512, seed=0)
    plan = driver.plan_genome(cfg, genome)
    assert len(plan[0]["tasks"]) == 512 and plan[0]["tasks"][0].num_lefs == 4979
    old = os.environ.get("MODLE_HIP_PAIRED")
    outs = {}
    try:
        for mode in (None, "0"):
            if mode is None:
                os.environ.pop("MODLE_HIP_PAIRED", None)
            else:
                os.environ["MODLE_HIP_PAIRED"] = mode
            sim = api.Simulator(cfg, 0)
            try:
                ids = driver.enqueue_plan(sim, cfg, plan)
                sim.launch()
                sim.wait()
                info = sim.launch_info()
                c, missed, occ = sim.copy_outputs(ids[0])
                res = sim.results(ids[0])
                summary = driver.verify_outputs(sim, cfg, plan, ids, [int(c.astype(np.int64).sum())], [missed],
                                                [int(occ.astype(np.int64).sum())])
                outs[mode] = (c, missed, occ, [(r.epochs, r.burnin_epochs, r.num_contacts, r.raws_consumed,
                                               list(r.prng_final)) for r in res], info, sim.kernel_ms())
            finally:
                sim.close()
            assert summary["tasks"] == 512 and summary["contacts"] > 0
    finally:
        if old is None:
            os.environ.pop("MODLE_HIP_PAIRED", None)
        else:
            os.environ["MODLE_HIP_PAIRED"] = old
    helper, single = outs[None], outs["0"]
    assert helper[4]["helper_waves"] == 1 and helper[4]["prng_producer_waves"] == 1 and helper[4]["main_waves_per_workgroup"] == 2
    assert single[4]["helper_waves"] == 0
    assert np.array_equal(helper[0], single[0]) and helper[1] == single[1] and np.array_equal(helper[2], single[2])
    assert helper[3] == single[3]
    # the launch lasts as long as its longest cell: the burn-in of some cell takes three times the mean
    epochs = [r[0] for r in helper[3]]
    assert max(epochs) > 2 * (sum(epochs) / len(epochs))
    print(f"chr1 x 512: helper-wave mode {helper[5]:.0f} ms, one wave per cell {single[5]:.0f} ms")


@pytest.mark.parametrize("skip_burnin,ncells,extra", [
    (0, 8, {}), (1, 6, {}),
    # BASELINE config 4 shape: 64 LEFs/Mb (15 933 LEFs on chr1), minor-collision trials, soft stalls
    (0, 3, dict(number_of_lefs_per_mbp=64.0, lef_bar_minor_collision_pblock=0.3,
                soft_stall_lef_stability_multiplier=2.0, num_cells=4096))])
def test_chr1_scale_matches_oracle(oracle, skip_burnin, ncells, extra):
    """BASELINE config 1 shape (chr1: 4979 LEFs, 3518 barriers) against the oracle, cell by cell.
    With burn-in skipped every LEF binds in epoch 0: the ranking sorts 4979 new keys in device
    memory instead of LDS, and barrier windows / staged slices are re-staged many times."""
    from modle_amd import api, synthetic
    from parity_cases import assert_same_outputs, assert_same_results

    chrom = synthetic.grch38_like(seed=42, chroms={"chr1"})[0]
    cfg = api.make_config(**dict(dict(num_cells=512, seed=0, skip_burnin=skip_burnin), **extra))
    tasks = api.slice_tasks(api.make_tasks(cfg, chrom["name"], chrom["size"], chrom["start"],
                                           chrom["end"]), 0, ncells)
    stp_a, stp_i = api.barrier_stps(cfg, chrom["bar_occupancy"])
    oc, om, oo, ores = oracle.simulate_interval(cfg, chrom["start"], chrom["end"], chrom["bar_pos"],
                                                chrom["bar_dir"], stp_a, stp_i, tasks, nthreads=8)
    sim = api.Simulator(cfg, 0)
    try:
        gc, gm, go, gres = sim.simulate_interval(chrom["start"], chrom["end"], chrom["bar_pos"],
                                                 chrom["bar_dir"], stp_a, stp_i, tasks)
    finally:
        sim.close()
    assert_same_results(ores, gres, "chr1")
    assert_same_outputs((oc, om, oo), (gc, gm, go), "chr1")


def test_interval_completion_counters():
    """modle_hip_interval_done: false for an interval whose cells are still running, true once
    they have all finished (and for every interval after wait)."""
    import time

    from modle_amd import api, driver, synthetic

    genome = [synthetic.synthetic_chromosome("chrBig", 60_000_000, seed=1),
              synthetic.synthetic_chromosome("chrSmall", 1_000_000, seed=2)]
    cfg = api.make_config(num_cells=4096, seed=3)
    plan = driver.plan_genome(cfg, genome)
    for e in plan:  # many long cells on the big interval, a few short ones on the small one
        e["tasks"] = api.slice_tasks(e["tasks"], 0, 3000 if e["interval"]["name"] == "chrBig" else 8)
    sim = api.Simulator(cfg, 0)
    try:
        ids = driver.enqueue_plan(sim, cfg, plan)
        assert all(sim.interval_done(i) for i in ids)  # nothing in flight
        sim.launch()
        seen_small_first = False
        t0 = time.time()
        while not sim.interval_done(ids[0]) and time.time() - t0 < 120:
            if sim.interval_done(ids[1]):
                seen_small_first = True
            time.sleep(0.001)
        sim.wait()
        assert all(sim.interval_done(i) for i in ids)
        # the 8 small cells finish long before the 3000 big ones (two rounds over 2048 waves)
        assert seen_small_first
        c_small = sim.copy_outputs(ids[1])[0]
        assert int(c_small.sum()) == sum(r.num_contacts for r in sim.results(ids[1]))
    finally:
        sim.close()


def test_bench_runs_under_torch_distributed_run_with_one_rank(tmp_path):
    """bench.py's own N > 1 code path (RCCL process group, per-interval reduce on a side stream as
    intervals complete, self-check) with one rank under torch.distributed.run."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
         "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(root, "bench.py"),
         "--gpus", "1", "--steps", "2", "--warmup", "1", "--cells", "16", "--no-cpu-baseline"],
        capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["checked"] is True and d["n_gpus"] == 1 and d["value"] > 0
    assert d["check"]["tasks"] == 24 * 16 and "side stream" in d["config"]["parallelism"]
    # the kernel time of every rank goes with the line (a real multi-GPU run shows its imbalance there)
    assert len(d["roofline"]["kernel_ms_per_rank"]) == 1 and d["roofline"]["kernel_ms_per_rank"][0] > 0
    # ... and the line says what kind of job it was: scaling mode, cells, the time the side stream spent in the
    # RCCL reduces, which GPU every rank ran on (distinct devices are asserted inside bench.py)
    mg = d["multi_gpu"]
    assert mg["scaling"] == "weak" and mg["total_cells"] == 16 and mg["cells_per_gpu"] == 16
    assert mg["backend"] == "RCCL" and mg["reduce_ms"] > 0 and len(mg["reduce_ms_per_rank"]) == 1
    assert len(mg["ranks"]) == 1 and mg["predicted"] is None  # (no prediction for one rank)


def _bench(tmp_path, nproc, port, extra):
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    sums = str(tmp_path / f"sums_{nproc}.json")
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += [os.path.join(root, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1",
            "--no-cpu-baseline", "--checksum-out", sums] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    with open(sums) as f:
        return json.loads(line), json.load(f)


def test_bench_with_two_real_ranks_reduces_to_the_single_rank_matrices(tmp_path):
    """bench.py's N > 1 path at world_size 2: two processes (fresh children of
    torch.distributed.run) share this GPU, every rank polls modle_hip_interval_done and issues the
    per-interval reduces in the same fixed order, the reduce itself goes through host copies (gloo:
    RCCL needs one GPU per rank).  With --scaling strong the job is the same 24 cells per
    chromosome as the single-process run, so the reduced matrices and occupancy tracks must be
    identical to it (sums and position-weighted sums of every interval)."""
    one, sums1 = _bench(tmp_path, 1, 0, ["--scaling", "strong", "--total-cells", "24"])
    two, sums2 = _bench(tmp_path, 2, 29531, ["--scaling", "strong", "--total-cells", "24",
                                            "--dist-backend", "gloo"])
    assert one["checked"] is True and two["checked"] is True
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["total_cells"] == one["config"]["total_cells"] == 24
    assert two["config"]["cells_per_gpu"] == 12 and "gloo" in two["config"]["parallelism"]
    assert len(sums1) == 24 and sums1 == sums2
    # whole-job counters are sums over the ranks
    assert two["check"]["tasks"] == 24 * 12  # rank 0's shard
    assert round(two["tasks_per_s"] * two["ms_per_step"] / 1e3) == 24 * 24
    assert len(two["roofline"]["kernel_ms_per_rank"]) == 2 and all(x > 0 for x in two["roofline"]["kernel_ms_per_rank"])
    assert one["roofline"]["kernel_ms_per_rank"] is None
    assert "multi_gpu" not in one
    mg = two["multi_gpu"]
    assert mg["scaling"] == "strong" and mg["total_cells"] == 24 and mg["cells_per_gpu"] == 12
    assert mg["backend"].startswith("gloo") and len(mg["reduce_ms_per_rank"]) == 2 and mg["reduce_ms"] > 0
    assert [r["device"] for r in mg["ranks"]] == [0, 0]  # (the rehearsal's ranks share this GPU: gloo only)
    assert mg["predicted"] is None  # (the committed prediction is for 16 384 cells, not for this job)

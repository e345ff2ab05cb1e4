"""The N > 1 path on CPU: two `gloo` ranks shard the cells of a small genome exactly like
bench.py / driver.py do on GPUs, sum-reduce their contact matrices and must reproduce the
single-rank result bit for bit (integer sums are order independent).  The per-rank compute is
done by the CPU oracle here -- the checker standing in for the GPU, which this container lacks;
what is under test is the sharding, task derivation and reduction logic."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _genome():
    from modle_amd import synthetic

    return [synthetic.synthetic_chromosome("chrA", 3_000_000, seed=1),
            synthetic.synthetic_chromosome("chrB", 2_000_000, seed=2, with_barriers=False),
            synthetic.synthetic_chromosome("chrC", 2_500_000, seed=3)]


def _simulate_shard(rank, world):
    from modle_amd import api, driver
    from oracle import binding as oracle

    cfg = api.make_config(num_cells=6, diagonal_width=1_000_000)
    plan = driver.plan_genome(cfg, _genome(), rank, world)
    outs = []
    for entry in plan:
        if entry["skipped"]:
            outs.append(None)
            continue
        iv = entry["interval"]
        stp_a, stp_i = api.barrier_stps(cfg, iv["bar_occupancy"])
        c, m, o, _ = oracle.simulate_interval(cfg, iv["start"], iv["end"], iv["bar_pos"],
                                              iv["bar_dir"], stp_a, stp_i, entry["tasks"])
        outs.append((c, m, o))
    return outs


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    outs = _simulate_shard(rank, world)
    for k, out in enumerate(outs):
        if out is None:
            continue
        c = torch.from_numpy(out[0].view(np.int32).copy())
        o = torch.from_numpy(out[2].view(np.int64).copy())
        m = torch.tensor([out[1]], dtype=torch.int64)
        for t in (c, o, m):
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            np.save(os.path.join(tmpdir, f"c{k}.npy"), c.numpy().view(np.uint32))
            np.save(os.path.join(tmpdir, f"o{k}.npy"), o.numpy().view(np.uint64))
            np.save(os.path.join(tmpdir, f"m{k}.npy"), m.numpy())
    dist.destroy_process_group()


def test_shard_bounds_partition_cells():
    from modle_amd import driver

    for n, w in ((512, 8), (10, 3), (5, 8), (2048, 1)):
        spans = [driver.shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_reduce_equals_single_rank(oracle, tmp_path):
    oracle.lib()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    single = _simulate_shard(0, 1)
    assert single[1] is None  # chrB has no barriers: skipped like the reference does
    for k, out in enumerate(single):
        if out is None:
            continue
        assert np.array_equal(np.load(tmp_path / f"c{k}.npy"), out[0])
        assert np.array_equal(np.load(tmp_path / f"o{k}.npy"), out[2])
        assert int(np.load(tmp_path / f"m{k}.npy")[0]) == out[1]


def _devices_worker(rank, world, port, tmpdir, shared):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from modle_amd import driver

    mine = ("host", 0 if shared else rank)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    try:
        driver.check_distinct_devices(everyone, world)
        verdict = "distinct"
    except RuntimeError as e:
        verdict = "shared: " + str(e)
    with open(os.path.join(tmpdir, f"devices_{rank}.txt"), "w") as f:
        f.write(verdict)
    dist.destroy_process_group()


@pytest.mark.parametrize("shared", [False, True])
def test_every_rank_reports_its_device_and_shared_gpus_are_refused(tmp_path, shared):
    """What bench.py does right after init_process_group for N > 1: the ranks
    all-gather (host, device) and every rank refuses a job in which two ranks sit on one GPU."""
    port = 31500 + os.getpid() % 2000 + (1 if shared else 0)
    mp.spawn(_devices_worker, args=(2, port, str(tmp_path), shared), nprocs=2, join=True)
    for r in range(2):
        verdict = (tmp_path / f"devices_{r}.txt").read_text()
        assert verdict.startswith("shared") if shared else verdict == "distinct"


def test_scaling_report_labels_the_job_and_quotes_the_prediction():
    import json

    from modle_amd import driver

    with open(os.path.join(ROOT, "profiles", "r05zs", "scale_prediction.json")) as f:
        pred = json.load(f)
    strong = driver.scaling_report(8, "strong", 16384, 2048, 11.5, pred)
    assert strong["scaling"] == "strong" and strong["total_cells"] == 16384 and strong["cells_per_gpu"] == 2048
    assert strong["reduce_ms"] == 11.5 and "strong" in strong["scaling_means"]
    assert 7.5 < strong["predicted"]["predicted_speedup"] < strong["predicted"]["predicted_speedup_kernel_only"] < 8
    assert "unmeasured" in strong["predicted"]["prediction_source"]
    # another job size: no prediction is quoted for it
    assert driver.scaling_report(8, "strong", 4096, 512, 1.0, pred)["predicted"] is None
    weak = driver.scaling_report(4, "weak", 8192, 2048, 9.0, pred)
    assert "weak" in weak["scaling_means"] and 3.9 < weak["predicted"]["predicted_speedup"] < 4.0
    assert driver.scaling_report(1, "weak", 2048, 2048, None, pred)["predicted"] is None
    with pytest.raises(RuntimeError):
        driver.check_distinct_devices([("h", 0), ("h", 0)], 2)
    assert driver.check_distinct_devices([("h", 0), ("h", 1)], 2)

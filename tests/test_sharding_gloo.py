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

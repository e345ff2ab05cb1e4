"""The whole-genome launches of BASELINE.json configs[2], [3] and [4] on the GPU.

These are the launches bench.py times (24 chromosomes, tens of thousands of (chromosome, cell)
tasks in ONE kernel launch); the oracle would need hours for them, so each config is covered twice:

* the full launch, checked through size-independent properties (driver.verify_outputs: every
  registered contact is a matrix increment or a missed update, per interval; every cell stops on
  its share of the target contacts -- the split of scheduler_simulate.cpp:129-141; the occupancy
  track is consistent with the sampling events; device status 0 for every task);
* a second launch of the same genome and Config with k cells per chromosome taken from the same
  task list (same PRNG states, same targets), compared with the oracle word for word: matrices,
  occupancy, missed updates and all per-cell counters including the final PRNG state.

configs[3] / [4] are 8-GPU configs: what runs here is the shard of rank 0 of 8 (cells
[0, num_cells / 8) of every chromosome), which is exactly what one GPU of the node executes.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIGS = {
    # configs[2]: whole GRCh38 + default barriers, 2048 cells, defaults, 1 GPU
    "config2_grch38_2048": dict(cfg=dict(num_cells=2048, seed=0), world=1),
    # configs[3]: 16384 cells over 8 GPUs -> rank 0 simulates cells [0, 2048)
    "config3_grch38_16384_rank0of8": dict(cfg=dict(num_cells=16384, seed=0), world=8),
    # configs[4]: 4096 cells, 64 LEFs/Mb, minor-collision trials + soft stalls, 8 GPUs -> 512 cells
    "config4_grch38_4096_dense_rank0of8": dict(
        cfg=dict(num_cells=4096, seed=0, number_of_lefs_per_mbp=64.0,
                 lef_bar_minor_collision_pblock=0.3, soft_stall_lef_stability_multiplier=2.0),
        world=8),
}


def _launch(cfg, plan):
    """one launch of a plan; returns (sim, ids, tensors, missed) -- caller closes sim"""
    import torch

    from modle_amd import api, driver

    dev = torch.device("cuda", 0)
    buffers, tensors = [], []
    for entry in plan:
        if entry["skipped"]:
            buffers.append((None, None))
            tensors.append(None)
            continue
        c = torch.zeros(entry["nrows"] * entry["ncols"] + 1, dtype=torch.int32, device=dev)
        o = torch.zeros(entry["ncols"], dtype=torch.int64, device=dev)
        buffers.append((c.data_ptr(), o.data_ptr()))
        tensors.append((c, o))
    sim = api.Simulator(cfg, 0)
    ids = driver.enqueue_plan(sim, cfg, plan, buffers)
    torch.cuda.synchronize(dev)
    sim.launch(torch.cuda.current_stream(dev).cuda_stream)
    sim.wait()  # raises if any task reports a non-zero device status
    return sim, ids, tensors, driver.read_missed(sim, ids)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_whole_genome_launch_properties(name):
    from modle_amd import api, driver, synthetic

    spec = CONFIGS[name]
    cfg = api.make_config(**spec["cfg"])
    genome = synthetic.grch38_like(seed=42)
    plan = driver.plan_genome(cfg, genome, 0, spec["world"])
    assert len(plan) == 24 and not any(e["skipped"] for e in plan)
    per_rank = int(cfg.num_cells) // spec["world"]
    assert all(len(e["tasks"]) == per_rank for e in plan)
    sim, ids, tensors, missed = _launch(cfg, plan)
    try:
        msum = [int(t[0].to(dtype=__import__("torch").int64).sum().item()) for t in tensors]
        osum = [int(t[1].sum().item()) for t in tensors]
        summary = driver.verify_outputs(sim, cfg, plan, ids, msum, missed, osum)
        assert summary["tasks"] == 24 * per_rank
        assert summary["contacts"] > 0
        assert sim.kernel_ms() > 0
        # the chromosomes are simulated to the same contact density: chr1's shard holds more
        # contacts than chrY's in proportion to its pixels (within the split's rounding)
        c1 = sum(t.num_target_contacts for t in plan[0]["tasks"])
        cy = sum(t.num_target_contacts for t in plan[23]["tasks"])
        assert msum[0] + missed[0] == c1 and msum[23] + missed[23] == cy and c1 > 4 * cy
    finally:
        sim.close()


@pytest.mark.parametrize("name,k", [("config2_grch38_2048", 2), ("config3_grch38_16384_rank0of8", 2),
                                    ("config4_grch38_4096_dense_rank0of8", 2)])
def test_whole_genome_sample_matches_oracle(oracle, name, k):
    from modle_amd import api, driver, synthetic
    from parity_cases import assert_same_outputs, assert_same_results

    spec = CONFIGS[name]
    cfg = api.make_config(**spec["cfg"])
    genome = synthetic.grch38_like(seed=42)
    plan = driver.plan_genome(cfg, genome, 0, spec["world"])
    # k cells of every chromosome out of the SAME task list (the last k of the shard: not the
    # cells every other test starts from)
    for e in plan:
        n = len(e["tasks"])
        e["tasks"] = api.slice_tasks(e["tasks"], n - k, n)
    # the oracle once, the launch in both modes (one wave per cell / main wave + helper: a launch
    # this small would otherwise always take the second)
    from parity_cases import launch_modes

    expected = []
    for entry in plan:
        iv = entry["interval"]
        stp_a, stp_i = api.barrier_stps(cfg, iv["bar_occupancy"])
        expected.append(oracle.simulate_interval(cfg, iv["start"], iv["end"], iv["bar_pos"],
                                                 iv["bar_dir"], stp_a, stp_i, entry["tasks"],
                                                 nthreads=k))
    for mode in launch_modes():
        sim, ids, tensors, missed = _launch(cfg, plan)
        try:
            for entry, iid, t, m, (oc, om, oo, ores) in zip(plan, ids, tensors, missed, expected):
                iv = entry["interval"]
                gc = t[0].cpu().numpy().view(np.uint32)
                go = t[1].cpu().numpy().view(np.uint64)
                what = f"{name}/{iv['name']}, helper waves {mode}"
                assert_same_results(ores, sim.results(iid), what)
                assert_same_outputs((oc, om, oo), (gc, m, go), what)
        finally:
            sim.close()

#!/usr/bin/env python3
"""Predicts the 1 -> 8 GPU curve of the strong-scaling job of BASELINE configs[3] on ONE GPU.

    python tools/scale_prediction.py OUT.json [--total-cells 16384] [--ranks 8 4 2]

`north_star` asks for ">= 6 x at 8 GPUs vs 1" on a job with a fixed number of cells (whole GRCh38,
16 384 cells: `bench.py --scaling strong --total-cells 16384`).  No multi-GPU node has been
available to this repository, so the N > 1 points are PREDICTED, not measured: cells are sharded
contiguously over ranks with no data-path collective (modle_amd/driver.py; reference:
scheduler_simulate.cpp:129-159, one queue of independent (interval, cell) tasks), so the time of an
N-GPU job is the time of its slowest shard plus the per-interval matrix reduce.  This script runs
the shard of EVERY rank of the 2-, 4- and 8-GPU jobs one after the other on the one GPU, and the
whole job once (the N = 1 leg), and records the kernel time of each launch (HIP events).

    predicted speed-up(N) = kernel_ms(N = 1) / max over ranks of kernel_ms(rank of N)

plus, per rank, the bytes of the final reduce (every interval's int32 band matrix and int64
occupancy track) and what they cost at one xGMI link's 153 GB/s.  What the prediction cannot see:
RCCL's own launch / synchronisation cost, link contention between the 7 simultaneous senders, and
host-side effects of 8 processes on one node.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

XGMI_LINK_GBS = 153.0  # per link and direction (MI355X_MICROARCH / task statement)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--total-cells", type=int, default=16384)
    ap.add_argument("--ranks", type=int, nargs="*", default=[8, 4, 2])
    ap.add_argument("--skip-single", action="store_true", help="do not run the N = 1 leg")
    args = ap.parse_args()

    from modle_amd import api, driver, synthetic

    genome = synthetic.grch38_like(seed=42)
    cfg = api.make_config(num_cells=args.total_cells, seed=0)

    def run_shard(rank, world):
        plan = driver.plan_genome(cfg, genome, rank, world)
        sim = api.Simulator(cfg, 0)
        try:
            sim.set_wait_timeout(900.0)
            ids = driver.enqueue_plan(sim, cfg, plan)
            t0 = time.perf_counter()
            sim.launch()
            sim.wait()  # raises unless every task reports status 0
            wall = time.perf_counter() - t0
            ms = sim.kernel_ms()
            epochs = 0
            for entry, iid in zip(plan, ids):
                if iid is not None:
                    epochs += sum(r.epochs for r in sim.results(iid))
            cells = len(plan[0]["tasks"])
        finally:
            sim.close()
        print(f"  world {world} rank {rank}: {cells} cells per chromosome, kernel {ms:.1f} ms, {epochs} cell-epochs",
              file=sys.stderr, flush=True)
        return {"rank": rank, "cells_per_chromosome": cells, "kernel_ms": ms, "wall_ms": wall * 1e3,
                "cell_epochs": epochs}

    reduce_bytes = 0
    for entry in driver.plan_genome(cfg, genome, 0, 1):
        reduce_bytes += (entry["nrows"] * entry["ncols"] + 1) * 4 + entry["ncols"] * 8
    out = {
        "what": "PREDICTED strong-scaling curve of BASELINE configs[3] (whole GRCh38-shaped genome, "
                f"{args.total_cells} cells, reference defaults, seed 0): every rank's shard run one after the "
                "other on ONE MI355X; unmeasured on multi-GPU hardware",
        "total_cells": args.total_cells,
        "reduce_bytes_per_rank": reduce_bytes,
        "reduce_ms_at_one_xgmi_link": reduce_bytes / (XGMI_LINK_GBS * 1e9) * 1e3,
        "worlds": {},
    }
    if not args.skip_single:
        print("N = 1 leg", file=sys.stderr, flush=True)
        out["single"] = run_shard(0, 1)
    for world in args.ranks:
        print(f"world {world}", file=sys.stderr, flush=True)
        shards = [run_shard(r, world) for r in range(world)]
        worst = max(s["kernel_ms"] for s in shards)
        entry = {"shards": shards, "max_kernel_ms": worst, "min_kernel_ms": min(s["kernel_ms"] for s in shards)}
        if "single" in out:
            entry["predicted_speedup_kernel_only"] = out["single"]["kernel_ms"] / worst
            entry["predicted_speedup_with_serial_reduce"] = out["single"]["kernel_ms"] / (
                worst + out["reduce_ms_at_one_xgmi_link"])
        out["worlds"][str(world)] = entry
        with open(args.out, "w") as f:  # (kept up to date: a long run leaves its progress behind)
            json.dump(out, f, indent=1)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "worlds"}
                     | {"speedups": {w: e.get("predicted_speedup_kernel_only") for w, e in out["worlds"].items()}}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Differential fuzz campaign on the GPU: python tools/fuzz_campaign.py FIRST_SEED N_SEEDS [v2|v3|v4]
(random set-ups of tests/fuzz_cases.py, HIP path vs oracle, all outputs and per-cell counters; every
set-up is launched in both modes: one wave per cell, and main wave + helper -- MODLE_HIP_PAIRED)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from fuzz_cases import random_case, random_case_v2, random_case_v3, random_case_v4  # noqa: E402
from modle_amd import api  # noqa: E402
from oracle import binding as oracle  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
gen = {"v2": random_case_v2, "v3": random_case_v3, "v4": random_case_v4}.get(sys.argv[3] if len(sys.argv) > 3 else "", random_case)
progress = open(os.environ["FUZZ_PROGRESS"], "w") if os.environ.get("FUZZ_PROGRESS") else None
bad = 0
skipped = 0
t0 = time.time()
for seed in range(first, first + count):
    case = gen(seed)
    cfg, chrom = case["cfg"], case["chrom"]
    tasks = api.slice_tasks(case["tasks"], 0, min(8, len(case["tasks"])))
    # keep the oracle's share of the run short: skip set-ups whose cells need many epochs
    per_epoch = max(1, api.compute_contacts_per_epoch(cfg, tasks[0].num_lefs))
    if cfg.target_contact_density >= 0 and tasks[0].num_target_contacts / per_epoch > 3000:
        skipped += 1
        continue
    track = bool(cfg.track_1d_lef_position)
    oc, om, oo, ores = oracle.simulate_interval(
        cfg, chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"],
        case["stp_active"], case["stp_inactive"], tasks, nthreads=8, track_occupancy=track)
    ok = True
    for mode in ("0", "1"):
        os.environ["MODLE_HIP_PAIRED"] = mode
        if progress is not None:
            # (what is about to be launched: a launch that hangs is then known by its seed and mode)
            progress.seek(0)
            progress.write(f"{seed} {mode} {case['kw']} size {case['size']}\n")
            progress.truncate()
            progress.flush()
        sim = api.Simulator(cfg, 0)
        gc, gm, go, gres = sim.simulate_interval(
            chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"], case["stp_active"],
            case["stp_inactive"], tasks)
        sim.close()
        ok = ok and np.array_equal(oc, gc) and om == gm and (not track or np.array_equal(oo, go))
        for a, b in zip(ores, gres):
            ok = ok and (a.epochs, a.burnin_epochs, a.num_contacts, a.raws_consumed, list(a.prng_final)) == (
                b.epochs, b.burnin_epochs, b.num_contacts, b.raws_consumed, list(b.prng_final))
    if (seed - first) % 25 == 24:
        print(f"  .. seed {seed}, {time.time() - t0:.0f} s", flush=True)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, case["kw"], case["size"], flush=True)
print(f"{count} seeds from {first}: {bad} mismatches, {skipped} skipped (long cells), {time.time() - t0:.0f} s")

#!/usr/bin/env python3
"""Differential fuzz campaign on the CPU lane emulator (tests/wave_emu: the product's device code,
64 lanes as fibers) against the oracle: python tools/emu_fuzz_campaign.py FIRST_SEED N_SEEDS
[v2|v3] [WORKERS].  One cell per set-up (the emulator is ~100x slower than the oracle); the way to
shake a change of the device code down before it goes to the GPU."""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(args):
    gen_name, seed = args
    import numpy as np
    import emu_sim
    import fuzz_cases
    from modle_amd import api
    from oracle import binding as oracle
    gen = {"v2": fuzz_cases.random_case_v2, "v3": fuzz_cases.random_case_v3,
           "v4": getattr(fuzz_cases, "random_case_v4", None)}.get(gen_name, fuzz_cases.random_case)
    case = gen(seed)
    cfg, chrom = case["cfg"], case["chrom"]
    tasks = api.slice_tasks(case["tasks"], 0, 1)
    per_epoch = max(1, api.compute_contacts_per_epoch(cfg, tasks[0].num_lefs))
    if cfg.target_contact_density >= 0 and tasks[0].num_target_contacts / per_epoch > 1500:
        return seed, None
    if tasks[0].num_lefs > 600:
        return seed, None  # (minutes under the emulator)
    track = bool(cfg.track_1d_lef_position)
    oc, om, oo, ores = oracle.simulate_interval(
        cfg, chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"],
        case["stp_active"], case["stp_inactive"], tasks, nthreads=1, track_occupancy=track)
    ec, em, eo, eres = emu_sim.simulate_interval(
        cfg, chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"], case["stp_active"],
        case["stp_inactive"], tasks, case["nrows"], case["ncols"], track_occupancy=track)
    ok = np.array_equal(oc, ec) and om == em and (not track or np.array_equal(oo, eo))
    a, b = ores[0], eres[0]
    ok = ok and (a.epochs, a.burnin_epochs, a.num_contacts, a.raws_consumed, list(a.prng_final)) == (
        b.epochs, b.burnin_epochs, b.num_contacts, b.raws_consumed, list(b.prng_final))
    return seed, bool(ok)


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    gen_name = sys.argv[3] if len(sys.argv) > 3 else "v1"
    workers = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    t0 = time.time()
    bad = skipped = 0
    with ProcessPoolExecutor(workers) as ex:
        for seed, ok in ex.map(run, [(gen_name, s) for s in range(first, first + count)], chunksize=4):
            if ok is None:
                skipped += 1
            elif not ok:
                bad += 1
                print("MISMATCH", gen_name, "seed", seed, flush=True)
    print(f"{gen_name}: {count} seeds from {first}: {bad} mismatches, {skipped} skipped, {time.time() - t0:.0f} s")

#!/usr/bin/env python3
"""Sums rocprofv3 --pmc outputs (counter_collection CSV files of separate passes) over the
launches of the simulation kernel and derives the figures committed under profiles/:

    python tools/pmc_summary.py OUT.json ALGORITHMIC_BYTES KERNEL_MS DIR [DIR ...]

HBM-side traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; FETCH_SIZE doubled on gfx950 as
MI355X_MICROARCH.md prescribes), per launch."""
import csv
import glob
import json
import os
import sys

KERNEL = "modle_simulate_cells"


def main():
    out, alg_bytes, kernel_ms = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
    totals, launches = {}, {}
    for d in sys.argv[4:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                seen = set()
                for row in csv.DictReader(f):
                    if KERNEL not in row.get("Kernel_Name", ""):
                        continue
                    name = row["Counter_Name"]
                    totals[name] = totals.get(name, 0.0) + float(row["Counter_Value"])
                    seen.add((name, row.get("Dispatch_Id")))
                for name, _ in seen:
                    launches[name] = launches.get(name, 0) + 1
    per_launch = {k: v / max(1, launches.get(k, 1)) for k, v in totals.items()}
    res = {"counters_per_launch": per_launch, "launches_seen": launches,
           "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kernel_ms}
    if "FETCH_SIZE" in per_launch and "WRITE_SIZE" in per_launch:
        fetch = per_launch["FETCH_SIZE"] * 1024.0
        write = per_launch["WRITE_SIZE"] * 1024.0
        traffic = 2.0 * fetch + write
        res.update(FETCH_SIZE_bytes_raw=fetch, WRITE_SIZE_bytes=write,
                   traffic_bytes_per_launch=traffic, traffic_over_algorithmic=traffic / alg_bytes,
                   traffic_GBps=traffic / (kernel_ms * 1e-3) / 1e9)
    if "TCC_HIT_sum" in per_launch:
        res["l2_hit_rate"] = per_launch["TCC_HIT_sum"] / (per_launch["TCC_HIT_sum"] + per_launch["TCC_MISS_sum"])
    if "SQ_WAVE_CYCLES" in per_launch:
        wc = per_launch["SQ_WAVE_CYCLES"]
        res["wave_cycles_waiting_fraction"] = per_launch.get("SQ_WAIT_ANY", 0.0) / wc
        res["wave_cycles_issuing_fraction"] = per_launch.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        res["wave_cycles_issue_stalled_fraction"] = per_launch.get("SQ_WAIT_INST_ANY", 0.0) / wc
    # the device sources the counters were taken on (bench.py quotes `traffic` only while they match)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.csrc_hash import csrc_sha256

    res["csrc_sha256"] = csrc_sha256(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

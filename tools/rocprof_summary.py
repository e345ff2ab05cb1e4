#!/usr/bin/env python3
"""Turns a rocprofv3 result (rocpd sqlite database, the default output of rocprofv3 7.x with
--kernel-trace --stats) into the per-kernel summary committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/<run>/prof profiles/r01_kernel_stats.csv
"""
import csv
import glob
import os
import sqlite3
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    dbs = sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))
    if not dbs:
        raise SystemExit(f"no rocpd database under {src}")
    rows = []
    for db in dbs:
        con = sqlite3.connect(db)
        cur = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
        rows += [(os.path.basename(db),) + r for r in cur.fetchall()]
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["database", "kernel", "calls", "total_us", "average_us", "percentage"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.3f}", f"{r[4]:.3f}", f"{r[5]:.4f}"])
    for r in rows[:6]:
        print(f"{r[1][:70]:70s} calls={r[2]} avg={r[4] / 1e3:.3f} ms ({r[5]:.2f} %)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""What a launch's tasks did in time: reads the per-task records of the profiling build
(MODLE_PROF_TASK_TIMES=<file> with libmodle_hip_prof.so: queue position, interval, start / end in
100 MHz ticks, wave slot, epochs, burn-in epochs) and prints, per interval, the spread of the cell
durations and epochs, the fit  time per epoch = a + b * LEFs,  and how far the launch's end is from
the mean of its waves (what the longest cells cost).

  MODLE_PROF_TASK_TIMES=gpurun_out/tasks.txt MODLE_HIP_LIB=libmodle_hip_prof.so python bench.py --steps 1 --warmup 0 --no-cpu-baseline
  python tools/task_timeline.py gpurun_out/tasks.txt[.gz] [LEFs per Mb, default 20]
"""
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    path = sys.argv[1]
    lefs_per_mb = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    from modle_amd import synthetic
    genome = synthetic.grch38_like(seed=42)
    sizes = np.array([c["end"] - c["start"] for c in genome], dtype=np.float64)
    a = np.loadtxt(gzip.open(path) if path.endswith(".gz") else open(path), dtype=np.int64)
    iv, t0, t1, slot, ep, bep = a.T
    base = t0.min()
    t0 = (t0 - base) * 1e-8
    t1 = (t1 - base) * 1e-8
    dur = t1 - t0
    out = {"tasks": int(len(a)), "waves": int(len(np.unique(slot))), "makespan_s": float(t1.max()),
           "queue_empty_s": float(t0.max()), "sum_task_s": float(dur.sum()),
           "mean_wave_busy_s": float(dur.sum() / len(np.unique(slot))),
           "burnin_share_of_epochs": float(bep.sum() / ep.sum()), "intervals": []}
    x, y = [], []
    for k in np.unique(iv):
        m = iv == k
        n_lefs = sizes[k] / 1e6 * lefs_per_mb if k < len(sizes) else float("nan")
        us_per_epoch = dur[m].sum() / ep[m].sum() * 1e6
        x.append(n_lefs)
        y.append(us_per_epoch)
        out["intervals"].append({
            "interval": int(k), "lefs": round(float(n_lefs)), "cells": int(m.sum()),
            "cell_s": [round(float(v), 3) for v in (dur[m].min(), dur[m].mean(), dur[m].max())],
            "epochs": [int(ep[m].min()), round(float(ep[m].mean())), int(ep[m].max())],
            "us_per_epoch": round(float(us_per_epoch), 1),
            "first_start_s": round(float(t0[m].min()), 2), "last_start_s": round(float(t0[m].max()), 2)})
    x, y = np.array(x), np.array(y)
    ok = np.isfinite(x)
    c = np.linalg.lstsq(np.vstack([np.ones(ok.sum()), x[ok]]).T, y[ok], rcond=None)[0]
    out["us_per_epoch_fit"] = {"intercept_us": round(float(c[0]), 1), "us_per_lef": round(float(c[1]), 4),
                               "us_per_256_lefs": round(float(c[1] * 256), 1)}
    late = np.argsort(-t1)[:8]
    out["last_to_finish"] = [{"queue_pos": int(i), "interval": int(iv[i]), "start_s": round(float(t0[i]), 3),
                              "cell_s": round(float(dur[i]), 3), "epochs": int(ep[i])} for i in late]
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

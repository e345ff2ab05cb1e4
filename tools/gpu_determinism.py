"""Diagnostic: many cells of a parity case on the GPU, twice; print checksums of every output."""
import hashlib, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from modle_amd import api
from parity_cases import build_case
name = sys.argv[1] if len(sys.argv) > 1 else "chr20mb_barriers"
ncell = int(sys.argv[2]) if len(sys.argv) > 2 else 512
case = build_case(name)
cfg, chrom = case["cfg"], case["chrom"]
tasks = api.slice_tasks(case["tasks"], 0, min(ncell, len(case["tasks"])))
for rep in range(2):
    sim = api.Simulator(cfg)
    t0 = time.time()
    c, missed, occ, res = sim.simulate_interval(0, chrom["size"], chrom["bar_pos"], chrom["bar_dir"], case["stp_active"], case["stp_inactive"], tasks)
    dt = time.time() - t0
    h = hashlib.sha1(c.tobytes()).hexdigest()[:12]
    ho = hashlib.sha1(occ.tobytes()).hexdigest()[:12]
    ep = sum(r.epochs for r in res); raws = sum(r.raws_consumed for r in res)
    print(name, len(tasks), "cells rep", rep, "contacts", h, "occ", ho, "epochs", ep, "raws", raws, "missed", missed, "%.2fs" % dt, flush=True)
    sim.close()

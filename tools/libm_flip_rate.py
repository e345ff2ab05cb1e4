#!/usr/bin/env python3
"""How often does a libm-dependent decision of the path come out differently between the shared
software log / exp / pow (modle_amd/csrc/modle_math.h: oracle AND device) and glibc's, which the
reference calls?  (DESIGN.md "Floating point".)

    python tools/libm_flip_rate.py FIRST_SEED N_SEEDS [v1|v2|v3|v4] [WORKERS]

Runs the first cells of every random set-up of tests/fuzz_cases.py through the two builds of the
oracle (oracle/libmodle_oracle.so and oracle/libmodle_oracle_libm.so, -DMO_USE_LIBM) and compares
the complete outcome of every cell: epochs, PRNG outputs drawn, final PRNG state, contact matrix.
A single flipped rejection test changes the number of outputs a draw consumes and with it the rest
of the cell's stream, so "cells that differ" counts cells with AT LEAST one flip; divided by the
outputs drawn it bounds the flip rate per draw from below by what was observed.  CPU only."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(first, count, gen_name):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fuzz_cases
    from modle_amd import api
    from oracle import binding as oracle

    gen = {"v2": fuzz_cases.random_case_v2, "v3": fuzz_cases.random_case_v3,
           "v4": fuzz_cases.random_case_v4}.get(gen_name, fuzz_cases.random_case)
    out = {}
    for seed in range(first, first + count):
        case = gen(seed)
        cfg, chrom = case["cfg"], case["chrom"]
        tasks = api.slice_tasks(case["tasks"], 0, min(4, len(case["tasks"])))
        per_epoch = max(1, api.compute_contacts_per_epoch(cfg, tasks[0].num_lefs))
        if cfg.target_contact_density >= 0 and tasks[0].num_target_contacts / per_epoch > 3000:
            continue
        c, m, o, res = oracle.simulate_interval(
            cfg, chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"], case["stp_active"],
            case["stp_inactive"], tasks, nthreads=1, track_occupancy=bool(cfg.track_1d_lef_position))
        out[str(seed)] = {"cells": [[r.epochs, r.burnin_epochs, r.num_contacts, r.raws_consumed,
                                      list(r.prng_final)] for r in res],
                          "matrix": hashlib.sha256(c.tobytes()).hexdigest()}
    json.dump(out, sys.stdout)


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    gen_name = sys.argv[3] if len(sys.argv) > 3 else "v1"
    workers = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libmodle_oracle.so",
                    "libmodle_oracle_libm.so"], check=True, capture_output=True)
    t0 = time.time()
    chunk = -(-count // workers)
    procs = []
    for lib in ("libmodle_oracle.so", "libmodle_oracle_libm.so"):
        for w in range(workers):
            a = first + w * chunk
            n = min(chunk, first + count - a)
            if n <= 0:
                continue
            env = dict(os.environ, MODLE_ORACLE_LIB=lib)
            procs.append((lib, subprocess.Popen([sys.executable, __file__, "--child", str(a), str(n), gen_name],
                                                stdout=subprocess.PIPE, env=env)))
    res = {"libmodle_oracle.so": {}, "libmodle_oracle_libm.so": {}}
    for lib, p in procs:
        data, _ = p.communicate()
        assert p.returncode == 0
        res[lib].update(json.loads(data))
    a, b = res["libmodle_oracle.so"], res["libmodle_oracle_libm.so"]
    assert a.keys() == b.keys()
    cells = differing = 0
    raws = 0
    bad_seeds = []
    for seed in a:
        for ca, cb in zip(a[seed]["cells"], b[seed]["cells"]):
            cells += 1
            raws += ca[3]
            if ca != cb:
                differing += 1
                if seed not in bad_seeds:
                    bad_seeds.append(seed)
        if a[seed]["matrix"] != b[seed]["matrix"] and seed not in bad_seeds:
            bad_seeds.append(seed)
    where = f" (seeds {bad_seeds})" if bad_seeds else ""
    print(f"{gen_name}: {len(a)} set-ups, {cells} cells, {raws} PRNG outputs drawn; "
          f"{differing} cells differ between modle_math.h and glibc{where}; "
          f"observed flips per output drawn: {differing / max(raws, 1):.3g}; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    else:
        main()

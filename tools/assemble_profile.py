#!/usr/bin/env python3
"""Turns what tools/profile_round.sh left under gpurun_out/<tag>/ into the committed evidence of a round:

    python tools/assemble_profile.py gpurun_out/r04zz profiles/r04z

copies the bench lines, the rocprofv3 kernel statistics and the phase breakdowns, sums the --pmc passes
(tools/pmc_summary.py), records the build's resource usage, and points profiles/traffic.json at the result
(with the hash of the device sources: bench.py prints the traffic only for the kernel it was measured on)."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for name in ("bench", "bench_under_rocprof", "bench_chr1_512", "bench_chr1_512_under_rocprof",
                 "bench_chr1_512_one_wave", "bench_no_tail_helpers", "bench_philox", "bench_grch38_dense"):
        shutil.copy(os.path.join(src, name + ".json"), dst)
    for trace, out in (("trace", "kernel_stats.csv"), ("trace_chr1", "kernel_stats_chr1_512.csv")):
        stats = glob.glob(os.path.join(src, trace, "**", "*_kernel_stats.csv"), recursive=True)
        shutil.copy(stats[0], os.path.join(dst, out))
    for name in ("phase_breakdown.txt", "phase_breakdown_dense.txt"):
        with open(os.path.join(src, name)) as f, open(os.path.join(dst, name), "w") as g:
            g.writelines(line for line in f if "amdgpu" not in line)
    summaries = {}
    for key, bench, out, dirs in (
            ("grch38:2048", "bench.json", "pmc_summary.json",
             ["pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_TCC_HIT_sum", "pmc_SQ_WAVE_CYCLES"]),
            ("grch38-dense:512", "bench_grch38_dense.json", "pmc_summary_dense.json",
             ["pmc_dense_FETCH_SIZE", "pmc_dense_WRITE_SIZE"])):
        roof = json.load(open(os.path.join(src, bench)))["roofline"]
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(dst, out),
                               str(roof["algorithmic_bytes_per_launch"]), str(roof["kernel_ms"])]
                              + [os.path.join(src, d) for d in dirs], stdout=subprocess.DEVNULL)
        summaries[key] = (out, json.load(open(os.path.join(dst, out))))
    traffic_path = os.path.join(ROOT, "profiles", "traffic.json")
    traffic = json.load(open(traffic_path))
    rel = os.path.relpath(dst, ROOT)
    for key, (out, s) in summaries.items():
        traffic[key].update(bytes_per_launch=s["traffic_bytes_per_launch"], csrc_sha256=s["csrc_sha256"],
                            kernel=f"kernel of {rel}",
                            source=f"{rel}/{out} (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes; "
                                   "2 x FETCH_SIZE + WRITE_SIZE, KiB counters)")
    json.dump(traffic, open(traffic_path, "w"), indent=1)
    # resource usage of the kernels of both size classes, as the library's own build recorded it (csrc/Makefile)
    shutil.copy(os.path.join(ROOT, "modle_amd", "csrc", "build_resources.txt"), os.path.join(dst, "build_resources.txt"))
    for name in sorted(os.listdir(dst)):
        if name.startswith("bench") and name.endswith(".json"):
            d = json.load(open(os.path.join(dst, name)))
            r = d["roofline"]
            print(f"{name:36s} {d['value']:8.1f} {d['unit']:16s} ms/step {d['ms_per_step']:8.1f} kernel {r['kernel_ms']:8.1f} "
                  f"frac {r['frac']:.4f} cpu {(d.get('cpu_baseline') or {}).get('value')}")
    for key, (out, s) in summaries.items():
        print(out, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in s.items()
                    if k not in ("counters_per_launch", "launches_seen")})


if __name__ == "__main__":
    main()

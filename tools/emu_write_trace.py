#!/usr/bin/env python3
"""Bytes of device memory each phase of the epoch loop CHANGES, per LEF and epoch, from the product's
device code on the CPU lane emulator (tests/wave_emu: `libmodle_emu_wtrace.so`, a build whose "clock"
snapshots the cell's workspace at the start of every phase and counts the 32-bit words that differ at its
end).  A lower bound of the bytes a phase writes (a value written again does not count; contact-matrix
increments are outside the workspace) -- the table VERDICT r03 1(b) asked for, by the one instrument this
repo has that sees every store: `rocprofv3 --pmc WRITE_SIZE` gives one number per launch.

    python tools/emu_write_trace.py [case ...]     (cases of tests/parity_cases.py; one cell each)
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PHASES = ["burnin_stats", "bind", "rank_rev", "rank_fwd", "sample", "gen_moves", "adjust_moves", "barriers+clear",
          "boundaries", "lef_bar", "primary", "secondary", "fix_secondary", "extrude_release", "lef_activation"]


def main():
    import emu_sim
    from modle_amd import api
    from parity_cases import build_case
    from phase_backend import emu_lib

    names = sys.argv[1:] or ["chr20mb_barriers", "many_rebinds_per_epoch", "dense_stress_rebinds_and_displaced"]
    # MODLE_WTRACE_SECTOR=64: the build that counts dirty 64-byte sectors (default: 32-byte sectors)
    variant = "wtrace64" if os.environ.get("MODLE_WTRACE_SECTOR", "32") == "64" else "wtrace"
    sector = 64 if variant == "wtrace64" else 32
    lib = emu_lib(variant)
    lib.emu_write_trace_read.argtypes = [C.POINTER(C.c_uint64 * 16), C.c_int]
    out = {}
    for name in names:
        case = build_case(name)
        cfg, chrom = case["cfg"], case["chrom"]
        tasks = api.slice_tasks(case["tasks"], 0, 1)
        buf = (C.c_uint64 * 16)()
        lib.emu_write_trace_read(C.byref(buf), 1)
        _, _, _, res = emu_sim.simulate_interval(
            cfg, chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"], case["stp_active"],
            case["stp_inactive"], tasks, case["nrows"], case["ncols"],
            track_occupancy=bool(cfg.track_1d_lef_position), variant=variant)
        lib.emu_write_trace_read(C.byref(buf), 1)
        lef_epochs = int(res[0].sum_active_lefs)
        # (low half of a reading: bytes changed; high half: dirty sectors -- tests/wave_emu/emu_api.cpp)
        lo = [buf[i] & 0xFFFFFFFF for i in range(15)]
        hi = [(buf[i] >> 32) * sector for i in range(15)]
        row = {PHASES[i]: round(lo[i] / lef_epochs, 2) for i in range(15)}
        row["all phases"] = round(sum(lo) / lef_epochs, 2)
        srow = {PHASES[i]: round(hi[i] / lef_epochs, 2) for i in range(15)}
        srow["all phases"] = round(sum(hi) / lef_epochs, 2)
        out[name] = {"lefs": int(tasks[0].num_lefs), "barriers": len(chrom["bar_pos"]), "epochs": int(res[0].epochs),
                     "burnin_epochs": int(res[0].burnin_epochs), "lef_epochs": lef_epochs,
                     "bytes_changed_per_lef_epoch": row,
                     f"bytes_in_dirty_{sector}B_sectors_per_lef_epoch": srow}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

"""Child process of the PHILOX-policy GPU tests: the product library is chosen when modle_amd is
imported (MODLE_HIP_LIB), so a run with the PHILOX build lives in its own interpreter.  Simulates
the first `ncells` cells of a parity case and stores outputs + per-cell counters in an .npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(case_name, ncells, out_path):
    assert os.environ.get("MODLE_HIP_LIB") == "libmodle_hip_philox.so"
    from modle_amd import api
    from parity_cases import build_case

    case = build_case(case_name)
    cfg, chrom = case["cfg"], case["chrom"]
    tasks = api.slice_tasks(case["tasks"], 0, min(int(ncells), len(case["tasks"])))
    sim = api.Simulator(cfg, 0)
    try:
        c, m, o, res = sim.simulate_interval(chrom["start"], chrom["end"], chrom["bar_pos"],
                                             chrom["bar_dir"], case["stp_active"],
                                             case["stp_inactive"], tasks)
    finally:
        sim.close()
    fields = ("epochs", "burnin_epochs", "num_contacts", "raws_consumed", "sum_active_lefs",
              "sampling_events", "sim_epochs")
    np.savez(out_path, contacts=c, missed=m, occupancy=o,
             results=np.array([[getattr(r, f) for f in fields] + list(r.prng_final) for r in res],
                              dtype=np.uint64))


if __name__ == "__main__":
    main(*sys.argv[1:4])

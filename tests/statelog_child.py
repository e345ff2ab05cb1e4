"""Child process of the internal-state-log GPU test (the recording build of the library is
chosen when modle_amd is imported): simulates a few cells and stores their records."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(case_name, ncells, cap, out_path):
    assert os.environ.get("MODLE_HIP_LIB") == "libmodle_hip_statelog.so"
    from modle_amd import api
    from parity_cases import build_case

    case = build_case(case_name)
    cfg, chrom = case["cfg"], case["chrom"]
    tasks = api.slice_tasks(case["tasks"], 0, int(ncells))
    sim = api.Simulator(cfg, 0)
    try:
        sim.enable_state_log(int(cap))
        iid = sim.add_interval(chrom["start"], chrom["end"], chrom["bar_pos"], chrom["bar_dir"],
                               case["stp_active"], case["stp_inactive"])
        sim.submit(iid, tasks)
        sim.launch()
        sim.wait()
        logs = [sim.state_log(iid, k) for k in range(len(tasks))]
        c, m, o = sim.copy_outputs(iid)
    finally:
        sim.close()
    np.savez(out_path, contacts=c, **{f"log{k}": l for k, l in enumerate(logs)})


if __name__ == "__main__":
    main(*sys.argv[1:5])

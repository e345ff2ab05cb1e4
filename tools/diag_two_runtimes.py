"""diagnostic: does the order in which torch and libmodle_hip.so initialise HIP matter on this box?"""
import subprocess
import sys

CASES = {
    "lib_dlopen_then_torch_then_create": """
from modle_amd import api
cfg = api.make_config(num_cells=4)
import torch
x = torch.zeros(4, device='cuda')
s = api.Simulator(cfg, 0); s.close(); print('ok')
""",
    "torch_first": """
import torch
x = torch.zeros(4, device='cuda')
from modle_amd import api
cfg = api.make_config(num_cells=4)
s = api.Simulator(cfg, 0); s.close(); print('ok')
""",
    "lib_create_then_torch": """
from modle_amd import api
cfg = api.make_config(num_cells=4)
s = api.Simulator(cfg, 0); s.close()
import torch
x = torch.zeros(4, device='cuda'); print('ok')
""",
}
for name, code in CASES.items():
    for rep in range(2):
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        last = (p.stdout.strip().splitlines() or ["-"])[-1]
        err = [l for l in p.stderr.splitlines() if "Error" in l or "error" in l][-1:] or [""]
        print(f"{name} #{rep}: rc={p.returncode} {last} {err[0][:150]}")
    maps = subprocess.run([sys.executable, "-c", code.replace("print('ok')", "print([l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l and 'r-xp' in l])")],
                          capture_output=True, text=True)
    print("   hip runtimes mapped:", maps.stdout.strip()[-300:])

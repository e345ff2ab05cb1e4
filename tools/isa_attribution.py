#!/usr/bin/env python3
"""Static attribution of the kernel's ISA to source functions.

    hipcc ... -S --offload-device-only -g1 -o /tmp/modle_hip_dev.s modle_hip.hip   (make -C modle_amd/csrc asm)
    python tools/isa_attribution.py /tmp/modle_hip_dev.s [--kernel modle_simulate_cells] [--lines FILE]

Every instruction of the kernel is charged to the source line of the last `.loc` in front of it
and, through a scan of the headers for function heads, to the function that line belongs to.  Per
function: instructions, vector / scalar / LDS / device-memory instructions, `v_readlane` /
`v_writelane` (what spilled scalar registers cost), `s_waitcnt`.  Static counts: a line inside a
loop counts once.  `--lines FILE` lists the hottest lines of one source file instead.
"""
import argparse
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "modle_amd", "csrc")

FUNC_HEAD = re.compile(
    r"^(?:template\s*<[^>]*>\s*)?(?:MODLE_DEV(?:_NOINLINE|_CALL|_MEMBER)?|__device__|__global__|static|inline)"
    r"[\w\s:<>\*&,]*?\b([A-Za-z_]\w*)\s*\(")


def function_table(path):
    """[(first line, name)] of the function heads of one source file (a heuristic scan)."""
    out = []
    try:
        with open(path) as f:
            lines = f.readlines()
    except OSError:
        return out
    for no, line in enumerate(lines, 1):
        m = FUNC_HEAD.match(line.strip()) if not line.startswith((" ", "\t")) or "MODLE_DEV_MEMBER" in line else None
        if m and not line.strip().endswith(";"):
            out.append((no, m.group(1)))
    return out


SPILL_VGPRS = set()


def classify(op, text=""):
    # reloads / saves of spilled scalar registers go through the VGPRs the prologue names
    # ("; implicit-def: $vgprN : SGPR spill to VGPR lane"); other v_readlane are broadcasts
    if op.startswith("v_readlane") and any(text.rstrip().endswith(f", v{r}, {ln}") or f", v{r}," in text for r in SPILL_VGPRS for ln in [""]):
        return "spill_rd"
    if op.startswith("v_writelane") and any(text.split()[1].rstrip(",") == f"v{r}" for r in SPILL_VGPRS):
        return "spill_wr"
    if op.startswith(("v_readlane", "v_readfirstlane")):
        return "readlane"
    if op.startswith("v_writelane"):
        return "writelane"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernel", default="modle_simulate_cells")
    ap.add_argument("--lines", default=None, help="list the hottest lines of this source file")
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--hot-depth", type=int, default=3,
                    help="loop depth from which an instruction counts as `hot` (task loop 1, epoch loop 2, sweeps 3)")
    ap.add_argument("--sort", default="all")
    ap.add_argument("--check-hot-scratch", action="store_true",
                    help="exit 1 when a vector register is reloaded from scratch inside a sweep")
    args = ap.parse_args()

    files = {}
    tables = {}
    per_func = collections.defaultdict(collections.Counter)
    per_line = collections.defaultdict(collections.Counter)
    inside = False
    ctx = "main"
    cur = ("?", 0)
    loc = re.compile(r"^\s*\.loc\s+(\d+)\s+(\d+)")
    filedir = re.compile(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?')
    spill_def = re.compile(r"implicit-def: \$vgpr(\d+) : SGPR spill to VGPR lane")
    # first pass: the kernel's text, its spill registers, and the loop depth of every instruction
    # (a backward branch to a label closes a loop over everything in between)
    body = []
    with open(args.asm) as f:
        grab = False
        for line in f:
            if not grab:
                if line.startswith("_Z") and args.kernel in line and line.split(";")[0].rstrip().endswith(":"):
                    grab = True
                continue
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                break
            body.append(line)
            m = spill_def.search(line)
            if m:
                SPILL_VGPRS.add(int(m.group(1)))
    label_at = {}
    for i, line in enumerate(body):
        t = line.split(";")[0].strip()
        if t.endswith(":") and t.startswith(".LBB"):
            label_at[t[:-1]] = i
    depth_delta = [0] * (len(body) + 1)
    for i, line in enumerate(body):
        t = line.strip().split()
        if len(t) >= 2 and (t[0].startswith("s_cbranch") or t[0] == "s_branch") and t[1] in label_at and label_at[t[1]] <= i:
            depth_delta[label_at[t[1]]] += 1
            depth_delta[i + 1] -= 1
    depth = []
    d = 0
    for i in range(len(body)):
        d += depth_delta[i]
        depth.append(d)
    body_index = -1
    scratch_hot = []
    with open(args.asm) as f:
        for line in f:
            m = filedir.match(line)
            if m:
                name = m.group(3) if m.group(3) else m.group(2)
                files[int(m.group(1))] = os.path.basename(name)
                continue
            if not inside:
                if line.startswith("_Z") and args.kernel in line and line.split(";")[0].rstrip().endswith(":"):
                    inside = True
                continue
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                break
            body_index += 1
            m = loc.match(line)
            if m:
                cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
                # the chain of inlined call sites in the comment says whose copy of the code this is
                ctx = "helper" if "sim_helper.h" in line else "main"
                continue
            s = line.strip()
            if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
                continue
            op = s.split()[0]
            kind = classify(op, s.split(";")[0])
            if op.startswith("scratch_load") and depth[body_index] >= args.hot_depth:
                # a VECTOR register reloaded from scratch inside a sweep: a memory round trip per block in
                # front of whatever uses it (round 4: two builds 2 % slower for four of these)
                scratch_hot.append(f"{cur[0]}:{cur[1]}  {s.split(';')[0].strip()}")
            if kind == "spill_rd" and depth[body_index] >= args.hot_depth:
                per_func_hot_key = True
            else:
                per_func_hot_key = False
            fname, lno = cur
            if fname not in tables:
                tables[fname] = function_table(os.path.join(CSRC, fname))
            func = "?"
            for first, name in tables[fname]:
                if first <= lno:
                    func = name
                else:
                    break
            key = f"{ctx[0]}|{fname}:{func}"
            per_func[key][kind] += 1
            per_func[key]["all"] += 1
            if per_func_hot_key:
                per_func[key]["spill_rd_hot"] += 1
            if depth[body_index] >= args.hot_depth:
                per_func[key]["hot"] += 1
            per_line[(fname, lno)][kind] += 1
            per_line[(fname, lno)]["all"] += 1

    kinds = ["all", "hot", "valu", "salu", "lds", "vmem", "readlane", "spill_rd", "spill_rd_hot", "spill_wr", "wait"]
    tot = collections.Counter()
    for c in per_func.values():
        tot.update(c)
    print("kernel", args.kernel, " ".join(f"{k}={tot[k]}" for k in kinds))
    print(f"vector-register reloads from scratch inside sweeps (loop depth >= {args.hot_depth}): {len(scratch_hot)}")
    for x in scratch_hot:
        print("   ", x)
    if args.check_hot_scratch:
        return 1 if scratch_hot else 0
    if args.lines:
        rows = [(c["all"], ln, c) for (fn, ln), c in per_line.items() if fn == args.lines]
        rows.sort(reverse=True)
        for n, ln, c in rows[:args.top]:
            print(f"{args.lines}:{ln:<6d}" + " ".join(f"{k}={c[k]:<6d}" for k in kinds))
        return
    rows = sorted(per_func.items(), key=lambda kv: -kv[1][args.sort])
    print(f"{'function':50s}" + "".join(f"{k:>9s}" for k in kinds))
    for key, c in rows[:args.top]:
        print(f"{key:50s}" + "".join(f"{c[k]:9d}" for k in kinds))


if __name__ == "__main__":
    sys.exit(main())

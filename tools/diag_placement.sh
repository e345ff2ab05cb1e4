# What differs between the two speeds a process can run at (tools/diag_tlb.sh: not the translation)?  The
# measurement build MODLE_EXP_REALLOC places the workspace anew at every launch, so ONE process shows both
# speeds; per launch: kernel time (HIP events) next to one set of rocprofv3 counters.
#   diag_placement.sh "<counter> <counter> ..."
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${DIAG_TAG:-placement}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export MODLE_HIP_LIB=libmodle_hip_exp_realloc.so MODLE_BENCH_TIMING=1
timeout -k 10 240 rocprofv3 --pmc $1 --output-format csv -d $O/run -- python3 $R/bench.py --steps 6 --warmup 0 --no-cpu-baseline > $O/run.json 2> $O/run.err
python3 - $O <<'PY'
import csv, glob, os, re, sys
O = sys.argv[1]
ms = [float(m) for m in re.findall(r"kernel ([0-9.]+) ms", open(os.path.join(O, "run.err")).read())]
rows = {}
for path in glob.glob(os.path.join(O, "run", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if "modle_simulate_cells" in row.get("Kernel_Name", ""):
            d = rows.setdefault(int(row["Dispatch_Id"]), {})
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for k, (disp, c) in enumerate(sorted(rows.items())):
    print(ms[k] if k < len(ms) else None, {n: f"{v:.5g}" for n, v in sorted(c.items())})
PY

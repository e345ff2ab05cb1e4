# Is the run-to-run spread of a box (two modes ~2 % apart, constant inside a process) the address
# translation?  Several processes, each: one step under rocprofv3 --pmc with the vector L1's translation
# counters; prints kernel time (HIP events) next to the UTCL1 misses / hits of that process.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${DIAG_TAG:-tlb}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/p$i.json 2> $O/p$i.err
  python3 - $O/p$i $O/p$i.json <<'PY'
import csv, glob, json, sys, os
d, j = sys.argv[1], sys.argv[2]
tot = {}
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if "modle_simulate_cells" in row.get("Kernel_Name", ""):
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
print(round(json.load(open(j))["roofline"]["kernel_ms"], 1), {k: f"{v:.4g}" for k, v in sorted(tot.items())})
PY
done

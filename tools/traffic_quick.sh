# HBM-side traffic of the simulation kernel on the quarter-size launch (512 cells per chromosome):
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (the TCC block cannot count both at once)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PROFILE_TAG:-traffic}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CELLS=${DIAG_CELLS:-512}
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$set -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --cells $CELLS > $O/pmc_$set.json 2> $O/pmc_$set.err
done
python3 - <<PY
import csv, glob, json
tot = {}
for path in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if "modle_simulate_cells" in row.get("Kernel_Name", ""):
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
d = json.load(open("$O/pmc_FETCH_SIZE.json"))
alg = d["roofline"]["algorithmic_bytes_per_launch"]
fetch, write = tot["FETCH_SIZE"] * 1024, tot["WRITE_SIZE"] * 1024
traffic = 2 * fetch + write
out = {"cells_per_chromosome": $CELLS, "kernel_ms": d["roofline"]["kernel_ms"], "algorithmic_bytes": alg,
       "FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write, "traffic_bytes": traffic,
       "traffic_over_algorithmic": traffic / alg}
json.dump(out, open("$O/traffic_quick.json", "w"), indent=1)
print(json.dumps(out))
PY

# Memory-side counters of several builds of the library on the quarter-size launch (512 cells per
# chromosome), one gpurun call:  exp_counters.sh <tag> libA.so libB.so ...
# Per build: kernel ms, WRITE_SIZE / FETCH_SIZE (KiB), write / read requests at the fabric side of L2
# (total and the 64-byte / 32-byte ones) and L1 -> L2 requests.  Separate --pmc passes (TCC slots).
set -e
R=$GRAFT_REPO_ROOT; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CELLS=${DIAG_CELLS:-512}
for lib in "$@"; do
  n=$(basename $lib .so)
  i=0
  for set in "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
             "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    MODLE_HIP_LIB=$lib rocprofv3 --pmc $set --output-format csv -d $O/${n}_pmc_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --cells $CELLS > $O/${n}_pmc_$i.json 2> $O/${n}_pmc_$i.err || echo "$n pass $i failed"
  done
  echo "$n counters done"
done
python3 - "$O" "$@" <<'PY'
import csv, glob, json, os, sys
O = sys.argv[1]
rows = {}
for lib in sys.argv[2:]:
    n = os.path.basename(lib)[:-3]
    tot = {}
    for path in glob.glob(f"{O}/{n}_pmc_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            if "modle_simulate_cells" in row.get("Kernel_Name", ""):
                tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    try:
        d = json.load(open(f"{O}/{n}_pmc_1.json"))
        tot["kernel_ms_under_pmc"] = d["roofline"]["kernel_ms"]
        tot["cell_epochs"] = d["config"]["cell_epochs_per_gpu_step"]
    except Exception as e:
        tot["error"] = str(e)
    rows[n] = tot
json.dump(rows, open(f"{O}/counters.json", "w"), indent=1)
keys = sorted({k for r in rows.values() for k in r})
print("counter".ljust(28) + "".join(n[-22:].rjust(24) for n in rows))
for k in keys:
    print(k.ljust(28) + "".join((f"{rows[n].get(k, float('nan')):.5g}" if not isinstance(rows[n].get(k), str) else "err").rjust(24) for n in rows))
PY

# instruction-fetch diagnostics of the simulation kernel (quarter-size launch)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PROFILE_TAG:-diag_ic}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CELLS=${DIAG_CELLS:-512}
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAIT_ANY" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --cells $CELLS > $O/pmc_$i.json 2> $O/pmc_$i.err || echo "pass $i failed"
  echo pmc pass $i done
done
python3 - <<PY
import csv, glob
tot = {}
for path in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if "modle_simulate_cells" in row.get("Kernel_Name", ""):
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
with open("$O/icache_counters.txt", "w") as f:
    for k in sorted(tot):
        f.write(f"{k} {tot[k]:.6g}\n")
        print(k, f"{tot[k]:.6g}")
PY

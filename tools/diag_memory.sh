# memory-system diagnostics of the simulation kernel (one gpurun call):
#   1. kernel time with 8 / 4 / 2 waves per CU pulling tasks (latency-bound or saturated?)
#   2. PMC passes on the fabric side of L2 (request sizes, stalls, queue levels) and on L1->L2 latency
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PROFILE_TAG:-diag}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CELLS=${DIAG_CELLS:-512}
for w in 8 4 2; do
  MODLE_HIP_ACTIVE_WAVES=$w python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --cells $CELLS > $O/waves_$w.json 2> $O/waves_$w.err
  python3 -c "import json;d=json.load(open('$O/waves_$w.json'));print('active waves $w: kernel', d['roofline']['kernel_ms'], 'ms')"
done
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_READ_sum TCC_WRITE_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --cells $CELLS > $O/pmc_$i.json 2> $O/pmc_$i.err || echo "pass $i failed"
  echo pmc pass $i done
done
python3 - <<PY
import csv, glob, os
tot = {}
for path in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if "modle_simulate_cells" in row.get("Kernel_Name", ""):
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
with open("$O/diag_counters.txt", "w") as f:
    for k in sorted(tot):
        f.write(f"{k} {tot[k]:.6g}\n")
        print(k, f"{tot[k]:.6g}")
PY

# LDS side of the simulation kernel (quarter-size launch): bank conflicts against all LDS cycles,
# issue stalls on the LDS pipe, busy cycles per instruction class
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PROFILE_TAG:-diag_lds}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CELLS=${DIAG_CELLS:-512}
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
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
with open("$O/lds_counters.txt", "w") as f:
    for k in sorted(tot):
        f.write(f"{k} {tot[k]:.6g}\n")
        print(k, f"{tot[k]:.6g}")
PY

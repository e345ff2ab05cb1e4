# what keeps the SIMDs busy: busy cycles per instruction class and the VALU instruction mix of the
# simulation kernel (quarter-size launch), at 8 and at 4 waves per CU
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PROFILE_TAG:-diag_valu}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CELLS=${DIAG_CELLS:-512}
for aw in 8 4; do
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  MODLE_HIP_ACTIVE_WAVES=$aw rocprofv3 --pmc $set --output-format csv -d $O/aw${aw}_pmc_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --cells $CELLS > $O/aw${aw}_pmc_$i.json 2> $O/aw${aw}_pmc_$i.err || echo "pass $i failed"
  echo aw $aw pmc pass $i done
done
python3 - <<PY
import csv, glob
tot = {}
for path in glob.glob("$O/aw${aw}_pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if "modle_simulate_cells" in row.get("Kernel_Name", ""):
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
with open("$O/valu_counters_aw${aw}.txt", "w") as f:
    for k in sorted(tot):
        f.write(f"{k} {tot[k]:.6g}\n")
        print($aw, k, f"{tot[k]:.6g}")
PY
done

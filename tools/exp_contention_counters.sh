# Round 5: memory latencies and wave states of the contention matrix (tools/exp_contention.sh), rocprofv3 --pmc
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05g; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for cfg in "256 8 chr21" "32 8 chr21" "256 1 chr21" "256 8 chr1" "32 8 chr1" "256 1 chr1"; do
  set -- $cfg; g=$1; w=$2; ch=$3; per=8; [ $ch = chr1 ] && per=2; cells=$((g * w * per)); n=${ch}_g${g}_w${w}
  i=0
  for set in "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
             "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
             "TCC_HIT_sum TCC_MISS_sum SQ_INSTS_LDS SQ_INSTS_VMEM"; do
    i=$((i+1))
    MODLE_HIP_GRID=$g MODLE_HIP_ACTIVE_WAVES=$w MODLE_HIP_TAIL_HELPERS=0 MODLE_HIP_PAIRED=0 rocprofv3 --pmc $set --output-format csv -d $O/${n}_pmc_$i -- \
      python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --chrom $ch --cells $cells > $O/${n}_pmc_$i.json 2> $O/${n}_pmc_$i.err || echo "$n pass $i failed"
  done
  echo "$n done"
done
python3 - "$O" <<'PY'
import csv, glob, json, os, sys
O = sys.argv[1]
names = ["chr21_g256_w8", "chr21_g32_w8", "chr21_g256_w1", "chr1_g256_w8", "chr1_g32_w8", "chr1_g256_w1"]
rows = {}
for n in names:
    tot = {}
    for path in glob.glob(f"{O}/{n}_pmc_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            if "modle_simulate_cells" in row.get("Kernel_Name", ""):
                tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    try:
        d = json.load(open(f"{O}/{n}_pmc_1.json"))
        tot["kernel_ms"] = d["roofline"]["kernel_ms"]
        tot["cell_epochs"] = d["config"]["cell_epochs_per_gpu_step"]
    except Exception as e:
        tot["error"] = str(e)
    def ratio(a, b, name):
        if tot.get(a) and tot.get(b): tot[name] = tot[a] / tot[b]
    ratio("TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum", "= fabric read latency (cycles)")
    ratio("TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_WRREQ_sum", "= fabric write latency (cycles)")
    ratio("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "= L1->L2 read latency (cycles)")
    ratio("TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum", "= L1->L2 write latency (cycles)")
    ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "= waves waiting")
    ratio("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "= waves issuing")
    ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "= waves stalled at issue")
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "TCP_TCC_READ_REQ_sum", "TCC_EA0_RDREQ_sum", "SQ_WAVE_CYCLES"):
        if tot.get(k) and tot.get("cell_epochs"): tot["per cell-epoch: " + k] = tot[k] / tot["cell_epochs"]
    rows[n] = tot
json.dump(rows, open(f"{O}/counters.json", "w"), indent=1)
keys = sorted({k for r in rows.values() for k in r if k.startswith("=") or k.startswith("per ") or k in ("kernel_ms", "cell_epochs")})
print("".ljust(44) + "".join(n.rjust(16) for n in rows))
for k in keys:
    print(k.ljust(44) + "".join((f"{rows[n].get(k, float('nan')):.5g}" if not isinstance(rows[n].get(k), str) else "err").rjust(16) for n in rows))
PY

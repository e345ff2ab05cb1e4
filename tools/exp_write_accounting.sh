# Round 5: where do the written bytes go?  Hardware counters on single-interval launches that
# the lane emulator's per-phase write trace (tools/emu_write_trace.py: bytes changed, and bytes of the 32- / 64-byte
# sectors that hold a changed byte) models one to one, with and without the output increments, in both size classes.
#   exp_write_accounting.sh <tag>
set -e
R=$GRAFT_REPO_ROOT; TAG=${1:-r05d}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() {  # name lib chrom cells env...
  n=$1; lib=$2; chrom=$3; cells=$4; shift 4
  i=0
  for set in "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
             "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    env MODLE_HIP_LIB=$lib MODLE_BENCH_NO_VERIFY=1 "$@" rocprofv3 --pmc $set --output-format csv -d $O/${n}_pmc_$i -- \
      python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --chrom $chrom --cells $cells > $O/${n}_pmc_$i.json 2> $O/${n}_pmc_$i.err || echo "$n pass $i failed"
  done
  echo "$n done"
}
run s20_narrow       libmodle_hip.so               synth:20000000 8192
run s20_wide         libmodle_hip.so               synth:20000000 8192 MODLE_HIP_SIZE_CLASS=wide
run s20_noatomics    libmodle_hip_exp_noatomics.so synth:20000000 8192
run c110_narrow      libmodle_hip.so               synth:110000000 4096
run c110_wide        libmodle_hip.so               synth:110000000 4096 MODLE_HIP_SIZE_CLASS=wide
run c110_noatomics   libmodle_hip_exp_noatomics.so synth:110000000 4096
python3 - "$O" <<'PY'
import csv, glob, json, os, sys
O = sys.argv[1]
rows = {}
for n in ["s20_narrow", "s20_wide", "s20_noatomics", "c110_narrow", "c110_wide", "c110_noatomics"]:
    tot = {}
    for path in glob.glob(f"{O}/{n}_pmc_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            if "modle_simulate_cells" in row.get("Kernel_Name", ""):
                tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    try:
        d = json.load(open(f"{O}/{n}_pmc_1.json"))
        tot["kernel_ms_under_pmc"] = d["roofline"]["kernel_ms"]
        tot["cell_epochs"] = d["config"]["cell_epochs_per_gpu_step"]
        tot["lef_epochs"] = d["config"]["lef_epochs_per_gpu_step"]
        tot["contacts+occupancy events"] = d["check"].get("contacts", 0) if isinstance(d.get("check"), dict) else 0
        tot["algorithmic_bytes"] = d["roofline"]["algorithmic_bytes_per_launch"]
    except Exception as e:
        tot["error"] = str(e)
    if "WRITE_SIZE" in tot and tot.get("lef_epochs"):
        tot["WRITE bytes per LEF-epoch"] = tot["WRITE_SIZE"] * 1024 / tot["lef_epochs"]
        tot["FETCH bytes per LEF-epoch (2 x FETCH_SIZE)"] = 2 * tot.get("FETCH_SIZE", 0) * 1024 / tot["lef_epochs"]
    rows[n] = tot
json.dump(rows, open(f"{O}/counters.json", "w"), indent=1)
keys = sorted({k for r in rows.values() for k in r})
print("counter".ljust(44) + "".join(n[-16:].rjust(18) for n in rows))
for k in keys:
    print(k.ljust(44) + "".join((f"{rows[n].get(k, float('nan')):.5g}" if not isinstance(rows[n].get(k), str) else "err").rjust(18) for n in rows))
PY

# Round 5: what do the waves of a launch share?  The chr21-shaped interval (934 LEFs) with 8 tasks per active wave,
# workgroups (= CUs in use) x waves per workgroup varied; one wave per cell, no helpers.  Per-wave rate =
# cell-epochs / (kernel s x waves).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R
echo "grid waves/WG  waves  kernel_ms  cell-epochs  per-wave-rate"
for cfg in "256 1" "256 2" "256 4" "256 6" "256 8" "128 8" "64 8" "32 8" "8 8" "32 1" "32 4" "8 1"; do
  set -- $cfg; g=$1; w=$2; cells=$((g * w * 8))
  MODLE_HIP_GRID=$g MODLE_HIP_ACTIVE_WAVES=$w MODLE_HIP_TAIL_HELPERS=0 MODLE_HIP_PAIRED=0 python3 bench.py --steps 2 --warmup 1 \
     --no-cpu-baseline --chrom chr21 --cells $cells > $O/g${g}_w${w}.json 2> $O/g${g}_w${w}.err || { echo "$cfg failed"; continue; }
  python3 - <<PY
import json
d=json.load(open("$O/g${g}_w${w}.json")); ms=d["roofline"]["kernel_ms"]; ce=d["config"]["cell_epochs_per_gpu_step"]
print("%4d %5d %7d %10.1f %12d %10.0f" % ($g, $w, $g*$w, ms, ce, ce/(ms/1e3)/($g*$w)))
PY
done

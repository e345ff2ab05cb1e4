# Round 5, step (i) of the LDS-resident kernel: what does residency of a cell's state in LDS buy per wave?
# Builds (make exp, see modle_hip.hip: MODLE_EXP_LDS_WS): the SAME waves of every workgroup run the SAME code
# with the unit arrays / barrier states / stalling-barrier lists in LDS (ldsws*) or in device memory (ldsoff*).
#   ldsws / ldsoff         2 waves per CU on ONE SIMD (waves 0 and 4), 19.2 Mb synthetic interval (384 LEFs, ~241 barriers)
#   ldsws1 / ldsoff1       2 waves per CU on two SIMDs (waves 0 and 1), same interval
#   ldsws_one / ldsoff_one 1 wave per CU, chr21-shaped interval (934 LEFs, 427 barriers)
# Reference points: the production library on the same workloads (8 waves per CU, loaded memory system).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
run() {  # tag lib chrom cells [env...]
  tag=$1; lib=$2; chrom=$3; cells=$4; shift 4
  env MODLE_HIP_LIB=$lib "$@" python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --chrom $chrom --cells $cells \
      --checksum-out $O/sum_$tag.json > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; tail -3 $O/$tag.err; return 1; }
  python3 - <<PY
import json
d=json.load(open("$O/$tag.json"))
r=d["roofline"]; c=d["config"]
print("%-22s kernel %9.1f ms  cell-epochs/launch %9d  waves/CU %s  cell-epochs per wave-second %9.1f" % (
  "$tag", r["kernel_ms"], c["cell_epochs_per_gpu_step"], "$tag".split("_w")[-1] if "_w" in "$tag" else "-",
  0.0))
PY
}
MODLE_HIP_LIB=libmodle_hip_exp_ldsws.so python3 __graft_entry__.py smoke 2>&1 | tail -1
S=synth:19200000
run small_lds_w2        libmodle_hip_exp_ldsws.so   $S 8192 &&
run small_hbm_w2        libmodle_hip_exp_ldsoff.so  $S 8192 &&
run small_lds2simd_w2   libmodle_hip_exp_ldsws1.so  $S 8192 &&
run small_hbm2simd_w2   libmodle_hip_exp_ldsoff1.so $S 8192 &&
run small_prod_w8       libmodle_hip.so             $S 32768 MODLE_HIP_TAIL_HELPERS=0 &&
run small_prodsame_w8   libmodle_hip.so             $S 8192 MODLE_HIP_TAIL_HELPERS=0 &&
run chr21_lds_w1        libmodle_hip_exp_ldsws_one.so  chr21 2048 &&
run chr21_hbm_w1        libmodle_hip_exp_ldsoff_one.so chr21 2048 &&
run chr21_prod_w8       libmodle_hip.so                chr21 16384 MODLE_HIP_TAIL_HELPERS=0 &&
run chr21_prodsame_w8   libmodle_hip.so                chr21 2048 MODLE_HIP_TAIL_HELPERS=0
cmp $O/sum_small_lds_w2.json $O/sum_small_hbm_w2.json && cmp $O/sum_small_lds_w2.json $O/sum_small_prodsame_w8.json && echo "checksums small: identical (LDS = HBM = production)"
cmp $O/sum_chr21_lds_w1.json $O/sum_chr21_hbm_w1.json && cmp $O/sum_chr21_lds_w1.json $O/sum_chr21_prodsame_w8.json && echo "checksums chr21: identical (LDS = HBM = production)"

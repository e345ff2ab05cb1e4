R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05place2; mkdir -p $O; cd $R
for a in ${CHUNKS:-malloc 2 32 256}; do
  echo "== alloc $a"
  MODLE_HIP_WORKSPACE_ALLOC=$a MODLE_HIP_EXP_PROBE=0 MODLE_HIP_LIB=libmodle_hip_exp_realloc.so MODLE_BENCH_TIMING=1 timeout -k 10 300 \
    python bench.py --steps 6 --warmup 0 --no-cpu-baseline > $O/$a.json 2> $O/$a.err
  grep -E "bench timing|failed" $O/$a.err | sed "s/.*(kernel/kernel/" | tr '\n' ' '; echo
done

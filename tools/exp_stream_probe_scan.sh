R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05probe; mkdir -p $O; cd $R
for u in 256 1024 4; do
MODLE_HIP_EXP_SCAN=$u MODLE_HIP_WORKSPACE_ALLOC=malloc MODLE_HIP_LIB=libmodle_hip_exp_realloc.so MODLE_BENCH_TIMING=1 timeout -k 10 300 \
  python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/scan$u.json 2> $O/scan$u.err
grep -E "scan:|bench timing" $O/scan$u.err | sed "s/.*(kernel/   kernel/; s/.*scan: //"
done

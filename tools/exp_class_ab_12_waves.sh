R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ab; mkdir -p $O; cd $R
MODLE_BENCH_ALTERNATE="MODLE_HIP_SIZE_CLASS=narrow,wide" MODLE_BENCH_TIMING=1 timeout -k 10 400 python bench.py --steps 6 --warmup 0 --no-cpu-baseline > $O/class.json 2> $O/class.err
grep "bench timing" $O/class.err | sed "s/.*(kernel/kernel/"

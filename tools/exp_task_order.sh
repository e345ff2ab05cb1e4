#!/bin/bash
# Large and small cells side by side instead of largest first (measurement build: make exp NAME=order FLAGS=-DMODLE_EXP_TASK_ORDER;
# MODLE_HIP_TASK_ORDER=0 sorted, 1 interleaved, 2 interleaved with the smallest fifth kept for the end), one process.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05order; mkdir -p $O; cd $R
MODLE_BENCH_ALTERNATE="MODLE_HIP_TASK_ORDER=0,1,2" MODLE_HIP_LIB=libmodle_hip_exp_order.so MODLE_BENCH_TIMING=1 timeout -k 10 400 \
  python bench.py --steps 9 --warmup 0 --no-cpu-baseline > $O/order.json 2> $O/order.err
grep "bench timing" $O/order.err | sed "s/.*(kernel/kernel/"

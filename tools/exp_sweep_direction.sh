#!/bin/bash
# Sweeps that run against the direction of the one before (the lines it left in the caches come first).  Measurement build
# make exp NAME=dir FLAGS=-DMODLE_EXP_SWITCH; MODLE_HIP_EXP bits switch a sweep back to its old direction; one process.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05dir; mkdir -p $O; cd $R
MODLE_BENCH_ALTERNATE="MODLE_HIP_EXP=${VALUES:-2,0}" MODLE_HIP_LIB=libmodle_hip_exp_dir.so MODLE_BENCH_TIMING=1 timeout -k 10 400 \
  python bench.py --steps ${STEPS:-8} --warmup 0 --no-cpu-baseline > $O/dir.json 2> $O/dir.err
grep "bench timing" $O/dir.err | sed "s/.*(kernel/kernel/"

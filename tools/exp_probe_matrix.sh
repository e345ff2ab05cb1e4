#!/bin/bash
# What about a placement of the workspace is slow?  The streaming probe on a fast and a slow allocation over the size of the hot
# set and the number of workgroups (measurement build: make exp NAME=pm FLAGS=-DMODLE_EXP_PROBE_MATRIX).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05pm; mkdir -p $O; cd $R
MODLE_HIP_LIB=libmodle_hip_exp_pm.so MODLE_BENCH_TIMING=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pm.json 2> $O/pm.err
grep -E "matrix|bench timing" $O/pm.err | sed "s/.*(kernel/   kernel/; s/.*matrix: //"

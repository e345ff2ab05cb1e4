#!/bin/bash
# The statistics' partition entries as 4 + 1 bytes (4 + 2 in the 8-wave kernels) instead of one 8-byte word: measurement build
# make exp NAME=pairs FLAGS=-DMODLE_EXP_SWITCH, MODLE_HIP_EXP=16 = the 8-byte entries; one process, launches alternate.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05pairs; mkdir -p $O; cd $R
MODLE_HIP_LIB=libmodle_hip_exp_pairs.so timeout -k 10 300 python -m pytest tests/test_size_classes.py tests/test_gpu_fuzz_parity.py -x -q -m gpu 2>&1 | tail -2
MODLE_BENCH_ALTERNATE="MODLE_HIP_EXP=16,0" MODLE_HIP_LIB=libmodle_hip_exp_pairs.so MODLE_BENCH_TIMING=1 timeout -k 10 400 \
  python bench.py --steps 8 --warmup 0 --no-cpu-baseline > $O/pairs.json 2> $O/pairs.err
grep "bench timing" $O/pairs.err | sed "s/.*(kernel/kernel/"

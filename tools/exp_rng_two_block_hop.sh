#!/bin/bash
# The 12-wave kernels' PRNG: one hop per PAIR of blocks (sim_types.h RNG_SPLIT) against one hop per block,
# both in one library (make exp NAME=rngsw FLAGS="-DMODLE_EXP_SWITCH -DMODLE_EXP_RNG_SWITCH"; MODLE_HIP_EXP=1 =
# the old scheme), alternating launch by launch inside one process (same workspace, same placement).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05rng; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_size_classes.py tests/test_gpu_fuzz_parity.py -x -q -m gpu > $O/pytest_classes_fuzz.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_classes_fuzz.txt
MODLE_BENCH_ALTERNATE="MODLE_HIP_EXP=1,0" MODLE_HIP_LIB=libmodle_hip_exp_rngsw.so MODLE_BENCH_TIMING=1 timeout -k 10 400 \
  python bench.py --steps 8 --warmup 0 --no-cpu-baseline > $O/ab.json 2> $O/ab.err
grep "bench timing" $O/ab.err | sed "s/.*(kernel/kernel/"

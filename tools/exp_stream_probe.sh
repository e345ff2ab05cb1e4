#!/bin/bash
# Does a streaming probe (every wave reads and writes through its own slot, no arithmetic) see which of the placement
# levels the next launch will run at?  Measurement build MODLE_EXP_REALLOC: workspace freed and allocated again behind a hole
# of varying size before every launch.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05probe; mkdir -p $O; cd $R
MODLE_HIP_WORKSPACE_ALLOC=${ALLOC:-malloc} MODLE_HIP_LIB=libmodle_hip_exp_realloc.so MODLE_BENCH_TIMING=1 timeout -k 10 500 \
  python bench.py --steps ${STEPS:-10} --warmup 0 --no-cpu-baseline > $O/probe.json 2> $O/probe.err
grep -E "stream probe|bench timing" $O/probe.err | sed "s/.*(kernel/   kernel/; s/.*workspace at/ws/"

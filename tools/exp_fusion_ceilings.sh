#!/bin/bash
# What are the loads worth that VERDICT r04's two fusions would remove?  Measurement build (make exp NAME=loads FLAGS=-DMODLE_EXP_SWITCH)
# in which a sweep can issue its block loads TWICE, the second set from cold arrays of the same slot (same bytes, same shape, real
# device-memory traffic): MODLE_HIP_EXP=4: the statistics' partition sweep (positions + ids of both directions = what "extrusion
# feeds the partition from registers" saves); 8: draw-free LEF-BAR detection (positions + moves = what "move adjustment + LEF-BAR
# in one pass" saves).  One process, launches alternate.  The cost of adding them bounds what removing them can give.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05fus; mkdir -p $O; cd $R
MODLE_BENCH_ALTERNATE="MODLE_HIP_EXP=0,4,8" MODLE_HIP_LIB=libmodle_hip_exp_loads.so MODLE_BENCH_TIMING=1 MODLE_BENCH_NO_VERIFY=1 timeout -k 10 500 \
  python bench.py --steps 9 --warmup 0 --no-cpu-baseline > $O/fus.json 2> $O/fus.err
grep "bench timing" $O/fus.err | sed "s/.*(kernel/kernel/"

#!/bin/bash
# Ways of drawing candidate allocations of the workspace (measurement build: make exp NAME=wss FLAGS=-DMODLE_EXP_WS_SEARCH):
# 0 = the loser freed, small holes of growing size kept (the product); 1 = every candidate held, no holes; 2 = held + holes;
# 3 = the loser freed, holes of odd MiB + a page.  24 draws each, the probe's ms per draw, and what 24 draws cost.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05wss; mkdir -p $O; cd $R
for rep in 1 2; do for s in 0 1 2 3; do
  MODLE_HIP_WS_STRATEGY=$s MODLE_HIP_LIB=libmodle_hip_exp_wss.so timeout -k 10 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/s$s.json 2> $O/s$s.err
  grep "ws search" $O/s$s.err | sed "s/.*exp\] //"
done; done

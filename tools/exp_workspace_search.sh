#!/bin/bash
# Fresh processes of the product with and without the workspace placement search (modle_hip.hip: place_workspace).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05search; mkdir -p $O; cd $R
for i in 1 2 3 4 5 6 7 8 9 10; do
  t=24; [ $((i % 3)) = 0 ] && t=1
  MODLE_HIP_WORKSPACE_TRIES=$t timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $O/p$i.json 2> $O/p$i.err
  python3 -c "
import json
d=json.load(open('$O/p$i.json'))
print('tries<=$t', round(d['value'],1), round(d['roofline']['kernel_ms'],1), d['config']['workspace_placement'])"
done

# Round 5: the two speeds of a launch and the placement of the workspace.  The measurement build MODLE_EXP_REALLOC frees the workspace
# and allocates it again behind a hole of 1 + 37 k mod 200 MiB at every launch (odd, even, odd, ... MiB); per launch: address, kernel ms.
#   alloc: hipMalloc or the virtual-memory-management API; probe: a kernel that reads one word per 4 KiB / 64 KiB / 2 MiB of the new buffer first
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05place; mkdir -p $O; cd $R
for cfg in "malloc 0" "vmm 0" "malloc 1" "vmm 1"; do
  set -- $cfg
  echo "== alloc $1, probe $2"
  MODLE_HIP_WORKSPACE_ALLOC=$1 MODLE_HIP_EXP_PROBE=$2 MODLE_HIP_LIB=libmodle_hip_exp_realloc.so MODLE_BENCH_TIMING=1 timeout -k 10 300 \
    python bench.py --steps 6 --warmup 0 --no-cpu-baseline > $O/$1_$2.json 2> $O/$1_$2.err
  grep -E "workspace at|bench timing" $O/$1_$2.err | sed "s/.*workspace at/ws/; s/.*(kernel/kernel/" | paste - - | cut -c1-110
done

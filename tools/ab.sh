# same-box A/B of two builds of the library: ab.sh <libA.so> <libB.so> [rounds] [extra bench args]
R=$GRAFT_REPO_ROOT; A=$1; B=$2; N=${3:-2}; shift 3 || true
for r in $(seq 1 $N); do for v in $A $B; do
  MODLE_HIP_LIB=$v python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/ab.json 2> $R/gpurun_out/ab.err
  python3 -c "import json;d=json.load(open('$R/gpurun_out/ab.json'));print('$v', round(d['roofline']['kernel_ms'],1), d['checked'])"
done; done

# same-box comparison of one library under two environments: ab_env.sh <rounds> "<ENV=..>" "<ENV=..>" [...]
# (an empty string stands for the default environment; bench arguments through AB_BENCH_ARGS as in ab.sh)
R=$GRAFT_REPO_ROOT; N=$1; shift
for r in $(seq 1 $N); do
  i=0
  for e in "$@"; do
    env $e python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $AB_BENCH_ARGS > $R/gpurun_out/ab.json 2> $R/gpurun_out/ab.err
    python3 -c "import json;d=json.load(open('$R/gpurun_out/ab.json'));print('[$e]', round(d['roofline']['kernel_ms'],1), d['checked'])"
  done
done

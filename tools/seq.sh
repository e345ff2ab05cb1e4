# runs bench.py once per library named on the command line, in that order (kernel time by HIP events):
# seq.sh libA.so libB.so libA.so ...   (how a build's time depends on what ran before it)
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  MODLE_HIP_LIB=$v python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $AB_BENCH_ARGS > $R/gpurun_out/ab.json 2> $R/gpurun_out/ab.err
  python3 -c "import json;d=json.load(open('$R/gpurun_out/ab.json'));print('$v', round(d['roofline']['kernel_ms'],1), d['checked'])"
done

# same-box comparison of builds of the library: ab.sh <rounds> <libA.so> <libB.so> [...]
# (file names inside modle_amd/; one bench.py run per build and round, kernel time from HIP events).
# The order of the builds is reversed in every other round: the second run of a pair tends to be a
# per cent or two slower than the first whatever it runs (clocks), which is more than most of the
# differences this script is asked about.
R=$GRAFT_REPO_ROOT; N=$1; shift
for r in $(seq 1 $N); do
  if [ $((r % 2)) -eq 1 ]; then order="$@"; else order=$(echo "$@" | tr ' ' '\n' | tac | tr '\n' ' '); fi
  for v in $order; do
    MODLE_HIP_LIB=$v python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $AB_BENCH_ARGS > $R/gpurun_out/ab.json 2> $R/gpurun_out/ab.err
    python3 -c "import json;d=json.load(open('$R/gpurun_out/ab.json'));print('$v', round(d['roofline']['kernel_ms'],1), d['checked'])"
  done
done

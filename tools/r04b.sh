set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
python3 bench.py --workload grch38-dense > $O/bench_grch38_dense.json 2> $O/bench_grch38_dense.err; cat $O/bench_grch38_dense.json
MODLE_HIP_LIB=libmodle_hip_prof.so python3 bench.py --workload grch38-dense --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_dense_prof.json 2> $O/phase_breakdown_dense.txt; grep -v amdgpu $O/phase_breakdown_dense.txt
MODLE_HIP_LIB=libmodle_hip_prof.so python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_prof.json 2> $O/phase_breakdown.txt; grep -v amdgpu $O/phase_breakdown.txt
python3 tools/scale_prediction.py $O/scale_prediction.json 2> $O/scale_prediction.err; tail -3 $O/scale_prediction.err

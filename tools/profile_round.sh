# One gpurun call that produces everything committed under profiles/<tag>/ for a round:
#   bench.json                 python bench.py (default workload, with the CPU baseline)
#   kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command (+ its bench line)
#   pmc_*                      separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / L2 hits / SQ)
#   phase_breakdown.txt        profiling build (per-phase wave time)
#   bench_chr1_512.json (+ kernel stats)   BASELINE config 1 (helper-wave mode), ..._one_wave.json without it
#   bench_no_tail_helpers.json             the default workload with MODLE_HIP_TAIL_HELPERS=0
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PROFILE_TAG:-r02}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
echo trace done
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_$tag.json 2> $O/pmc_$tag.err
  echo pmc $tag done
done
MODLE_HIP_LIB=libmodle_hip_prof.so python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_prof.json 2> $O/phase_breakdown.txt
grep -v amdgpu $O/phase_breakdown.txt
python3 $R/bench.py --workload chr1 --cells 512 > $O/bench_chr1_512.json 2> $O/bench_chr1_512.err; cat $O/bench_chr1_512.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_chr1 -- python3 $R/bench.py --workload chr1 --cells 512 --no-cpu-baseline > $O/bench_chr1_512_under_rocprof.json 2> $O/rocprof_chr1.err
# the same launch with one wave per cell (helper waves off), and the default launch without tail helpers
MODLE_HIP_PAIRED=0 MODLE_HIP_TAIL_HELPERS=0 python3 $R/bench.py --workload chr1 --cells 512 --no-cpu-baseline > $O/bench_chr1_512_one_wave.json 2> $O/bench_chr1_512_one_wave.err; cat $O/bench_chr1_512_one_wave.json
MODLE_HIP_TAIL_HELPERS=0 python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_no_tail_helpers.json 2> $O/bench_no_tail_helpers.err; cat $O/bench_no_tail_helpers.json
python3 $R/bench.py --rng philox --no-cpu-baseline > $O/bench_philox.json 2> $O/bench_philox.err; cat $O/bench_philox.json
ls -R $O | head -60
# BASELINE configs[4] (rank 0 of 8 of the collision-heavy stress): bench line with CPU baseline, phase breakdown, traffic
python3 $R/bench.py --workload grch38-dense > $O/bench_grch38_dense.json 2> $O/bench_grch38_dense.err; cat $O/bench_grch38_dense.json
MODLE_HIP_LIB=libmodle_hip_prof.so python3 $R/bench.py --workload grch38-dense --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_dense_prof.json 2> $O/phase_breakdown_dense.txt
grep -v amdgpu $O/phase_breakdown_dense.txt
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_dense_$set -- python3 $R/bench.py --workload grch38-dense --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_dense_$set.json 2> $O/pmc_dense_$set.err
  echo pmc dense $set done
done
ls -R $O | head -80

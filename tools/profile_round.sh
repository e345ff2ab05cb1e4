set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PROFILE_TAG:-r01g}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
echo trace done
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_$tag.json 2> $O/pmc_$tag.err
  echo pmc $tag done
done
ls -R $O | head -50

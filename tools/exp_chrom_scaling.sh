set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02d; mkdir -p $O
for c in chr1 chr8 chr16 chr21; do
  python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --chrom $c --cells 4096 > $O/only_$c.json 2> $O/only_$c.err
  python3 -c "
import json;d=json.load(open('$O/only_$c.json'));r=d['roofline'];e=d['config']['cell_epochs_per_gpu_step']
print('$c', 'kernel_ms', round(r['kernel_ms'],1), 'cell_epochs', e, 'alg_bytes', r['algorithmic_bytes_per_launch'], 'GB/s', round(r['achieved'],1))"
done

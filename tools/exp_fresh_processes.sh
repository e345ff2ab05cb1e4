# Round 5: what are the 12-wave kernels worth in a process of their own?  Four fresh processes with the library's
# choice (12 waves per workgroup on the default launch) and two with 8 forced; kernel ms by HIP events.
# (profiles/r05zz/waves_8_vs_12_fresh_processes.txt: the answer depends on the box.)
for i in 1 2 3 4; do python bench.py --steps 2 --warmup 0 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fresh 12:', [round(x) for x in [d['roofline']['kernel_ms']]], d['config']['waves_per_workgroup'])"; done
for i in 1 2; do MODLE_HIP_WAVES=8 python bench.py --steps 2 --warmup 0 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fresh  8:', [round(x) for x in [d['roofline']['kernel_ms']]], d['config']['waves_per_workgroup'])"; done

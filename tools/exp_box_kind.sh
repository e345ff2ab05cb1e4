# Round 5: what distinguishes the boxes on which the 12-wave kernels are worth 6 % from those where they are worth 1 %?
# (all of them: SPX / NPS1, 1 400 W cap, fclk 1 250 MHz, mclk 2 000 MHz, 2.4 GHz)  Temperatures and power while the kernels run.
sample() { while true; do rocm-smi --showtemp --showpower 2>/dev/null | grep -E "Temperature|Power \(W\)" | sed 's/.*GPU\[0\][^:]*: //' | tr '\n' ';'; echo; sleep 2; done; }
for w in 12 8; do
  sample > /tmp/smi_$w.txt & S=$!
  MODLE_HIP_WAVES=$w python bench.py --steps 3 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves $w: kernel', round(d['roofline']['kernel_ms']), 'ms')"
  kill $S
  tail -3 /tmp/smi_$w.txt | head -2
done

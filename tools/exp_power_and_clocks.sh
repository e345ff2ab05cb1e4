# Round 5: do the boxes differ in what the 12-wave kernels are worth because of a power cap?  Samples clocks and power
# while a fresh process runs the default launch with 12 and with 8 waves per workgroup.
rocm-smi --showmaxpower --showpower 2>/dev/null | grep -E "Power" | head -4
sample() { while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 1; done; }
for w in 12 8; do
  sample > /tmp/smi_$w.txt & S=$!
  MODLE_HIP_WAVES=$w python bench.py --steps 3 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves $w: kernel', round(d['roofline']['kernel_ms']), 'ms')"
  kill $S
  echo "  samples under load (sclk, W):"; grep -E "\(2[0-9]{3}Mhz\)|\(1[0-9]{3}Mhz\)" /tmp/smi_$w.txt | awk '{print $0}' | sort | uniq -c | sort -rn | head -6
done

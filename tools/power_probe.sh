#!/bin/bash
# Samples board power / clocks with rocm-smi while bench.py runs (diagnostic): prints the samples with the highest power.
python bench.py --steps 400 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/power_bench.log 2>&1 &
BP=$!
: > gpurun_out/power_samples.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' ' >> gpurun_out/power_samples.txt
  echo >> gpurun_out/power_samples.txt
  sleep 0.2
done
sort -t' ' -k3 -n -r gpurun_out/power_samples.txt | head -8
tail -1 gpurun_out/power_bench.log | cut -c100-200

#!/bin/bash
# Evidence for "the MFMA-bound layers are clock / power bound" (DESIGN.md section 6): in-kernel clock of the ping-pong kernel from
# s_memtime / s_memrealtime stamps on three layers, and rocm-smi socket power / sclk sampled while bench.py runs.
# usage (GPU box): bash tools/clock_power_evidence.sh > profiles/<round>_mfma_clock_power.txt
echo "== in-kernel stamps, conv_pingpong_kernel stamp build (tools/pp_stamps.py): cycles per K tile (2048 = back-to-back MFMA issue) and clock"
for shape in "200 336 256 256 3" "50 84 1024 256 1" "100 168 512 256 1"; do
  echo "-- layer H W Cin Cout k = $shape"
  python tools/pp_stamps.py $shape 2>/dev/null | head -1
done
echo "== rocm-smi while bench.py runs (socket power W, sclk): ten highest-power samples of a 0.2 s poll"
python bench.py --steps 600 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/power_bench.log 2>&1 &
BP=$!
: > /tmp/power_samples.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' ' | awk '{print $NF, $0}' >> /tmp/power_samples.txt
  sleep 0.2
done
sort -g -r /tmp/power_samples.txt | head -10 | cut -d' ' -f2-
echo "samples: $(wc -l < /tmp/power_samples.txt)"
echo "== idle sample"
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"
echo "== bench line of that run"
tail -1 /tmp/power_bench.log | cut -c1-260

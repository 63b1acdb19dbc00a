#!/bin/bash
# does the timed region depend on how long the GPU has been busy before it?  bench.py with different warmup lengths, same box, separate processes, twice
pick() { grep -o "\"steps\": [0-9]*, \"warmup\": [0-9]*, \"ms_per_step\": [0-9.]*"; }
for r in 1 2; do
  for W in 2 5 20 60 150; do
    python bench.py --steps 10 --warmup $W --no-cpu-baseline --no-roofline --no-from-uint8 2>/dev/null | pick
  done
done

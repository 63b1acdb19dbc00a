#!/bin/bash
# images/s against the per-GPU batch (SURVEY 8d lists B in {1, 2, 4, 8} for configs[2]; bench.py's default is 120): eager one stream, eager two streams
# (even batches), and the HIP-graph replay for the small ones.  usage: bash tools/batch_sweep.sh [config]
CFG=${1:-configs/faster_rcnn/faster_rcnn_r50_fpn.py}
pick() { grep -o "\"value\": [0-9.]*, \"unit\": \"images/sec\", \"n_gpus\": 1, \"steps\": [0-9]*, \"warmup\": [0-9]*, \"ms_per_step\": [0-9.]*" | sed 's/"unit": "images\/sec", "n_gpus": 1, //'; }
for B in 1 2 4 8 16 32 60 120; do
  S=$((B < 8 ? 40 : (B < 60 ? 20 : 10)))
  A=$(timeout -k 10 300 python bench.py --config $CFG --batch $B --steps $S --warmup 3 --no-cpu-baseline --no-roofline --no-from-uint8 --streams 1 2>/dev/null | pick)
  echo "b$B one stream: $A"
  if [ $((B % 2)) -eq 0 ]; then
    A=$(timeout -k 10 300 python bench.py --config $CFG --batch $B --steps $S --warmup 3 --no-cpu-baseline --no-roofline --no-from-uint8 --streams 2 2>/dev/null | pick)
    echo "b$B two streams: $A"
  fi
  if [ $B -le 16 ]; then
    A=$(timeout -k 10 300 python bench.py --config $CFG --batch $B --steps $S --warmup 3 --no-cpu-baseline --no-from-uint8 --streams 1 --graph 2>/dev/null | pick)
    echo "b$B graph replay: $A"
  fi
done

# same-box A/B of whole bench.py runs: persistent ping-pong form for K >= 1 (all eligible layers) / 2304 (default) / off, interleaved
CFG=${1:-configs/faster_rcnn/faster_rcnn_r50_fpn.py}; B=${2:-120}; R=${3:-3}
for r in $(seq $R); do
  for k in 1 2304 100000000; do
    MD_PERS_MIN_K=$k timeout -k 10 300 python bench.py --config $CFG --batch $B --steps 10 --no-cpu-baseline --no-roofline --no-from-uint8 2>/dev/null \
      | grep -o "\"value\": [0-9.]*, \"unit\": \"images/sec\", \"n_gpus\": 1, \"steps\": 10, \"warmup\": 3, \"ms_per_step\": [0-9.]*" | sed "s|^|$CFG b$B pers_min_k $k: |"
  done
done

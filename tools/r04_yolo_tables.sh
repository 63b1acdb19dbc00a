#!/bin/bash
# per-layer tables of the YOLO shards at batch 32 AND batch 128 (is TFLOP/s grid-bound or are the small-channel kernels slow?)
set -o pipefail
OUT=gpurun_out/r04_yolo
mkdir -p $OUT
for cfg in "yolov5s configs/yolov5/yolov5s.py" "yolov8l configs/yolov8/yolov8l.py"; do
  set -- $cfg
  for b in 32 128; do
    python bench.py --config $2 --batch $b --steps 10 --warmup 3 --no-cpu-baseline --bracket all --streams 1 --dump-convs $OUT/$1_b${b}_conv_layers.json > $OUT/$1_b${b}_bench.json 2> $OUT/$1_b${b}_bench.err || exit 1
    tail -c 300 $OUT/$1_b${b}_bench.json; echo
  done
done

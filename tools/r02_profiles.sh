#!/bin/bash
# round-2 evidence run on ONE MI355X: bench lines + per-layer tables + rocprofv3 kernel stats for the three single-GPU configs.
# usage (from the repo root on the GPU box): bash tools/r02_profiles.sh <tag>
set -e -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run_cfg() {  # name config batch
  python bench.py --config $2 --batch $3 --steps 10 --warmup 3 --no-cpu-baseline --bracket all --dump-convs $OUT/$1_conv_layers.json > $OUT/$1_bench.json 2> $OUT/$1_bench.err
  tail -c 600 $OUT/$1_bench.json; echo
}
python bench.py --steps 20 --warmup 5 > $OUT/frcnn_bench.json 2> $OUT/frcnn_bench.err
tail -c 1500 $OUT/frcnn_bench.json; echo
run_cfg frcnn_all configs/faster_rcnn/faster_rcnn_r50_fpn.py 120
run_cfg yolov5s configs/yolov5/yolov5s.py 32
run_cfg yolov8l configs/yolov8/yolov8l.py 32
for c in "frcnn configs/faster_rcnn/faster_rcnn_r50_fpn.py 120" "yolov5s configs/yolov5/yolov5s.py 32" "yolov8l configs/yolov8/yolov8l.py 32"; do
  set -- $c
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_$1 -- python3 $ROOT/bench.py --config $ROOT/$2 --batch $3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-zero-operands --no-from-uint8 > $ROOT/$OUT/prof_$1.log 2>&1)
  f=$(ls $OUT/prof_$1/*/*kernel_stats.csv | head -1)
  cp $f $OUT/$1_kernel_stats.csv
  rm -rf $OUT/prof_$1
  head -12 $OUT/$1_kernel_stats.csv
done

#!/bin/bash
# round-4 evidence run on ONE MI355X: bench lines, per-layer tables, rocprofv3 kernel stats, PMC traffic for the single-GPU configs.
# usage (from the repo root on the GPU box): bash tools/r04_profiles.sh <tag>      -> gpurun_out/<tag>/
set -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run_cfg() {  # name config batch [extra bench args]
  n=$1; c=$2; b=$3; shift 3
  python bench.py --config $c --batch $b --steps 10 --warmup 3 --no-cpu-baseline --bracket all --dump-convs $OUT/${n}_conv_layers.json "$@" > $OUT/${n}_bench.json 2> $OUT/${n}_bench.err
  tail -c 400 $OUT/${n}_bench.json; echo
}
echo "== PMC traffic first (bench.py reports it only when the file matches the kernel sources)"
bash tools/pmc_conv_traffic.sh $OUT/frcnn_conv_traffic.json 120 --streams 1 && cp $OUT/frcnn_conv_traffic.json profiles/r04_conv_traffic.json
bash tools/pmc_conv_traffic.sh $OUT/yolov5s_conv_traffic.json 32 --config configs/yolov5/yolov5s.py && cp $OUT/yolov5s_conv_traffic.json profiles/r04_yolov5s_conv_traffic.json
bash tools/pmc_conv_traffic.sh $OUT/yolov8l_conv_traffic.json 32 --config configs/yolov8/yolov8l.py --streams 1 && cp $OUT/yolov8l_conv_traffic.json profiles/r04_yolov8l_conv_traffic.json
bash tools/pmc_conv_traffic.sh $OUT/maskrcnn_conv_traffic.json 32 --config configs/mask_rcnn/mask_rcnn_r101_fpn.py --streams 1 && cp $OUT/maskrcnn_conv_traffic.json profiles/r04_maskrcnn_conv_traffic.json
echo "== default bench line"
python bench.py --steps 20 --warmup 5 > $OUT/frcnn_bench.json 2> $OUT/frcnn_bench.err
tail -c 1200 $OUT/frcnn_bench.json; echo
run_cfg frcnn_all configs/faster_rcnn/faster_rcnn_r50_fpn.py 120 --streams 1
run_cfg yolov5s configs/yolov5/yolov5s.py 32
# the configs that run two HIP streams (test_cfg.streams = 2): the per-layer table comes from a one-stream run (--bracket all needs it), the bench line
# from the config as it ships
run_cfg yolov8l_one_stream configs/yolov8/yolov8l.py 32 --streams 1
run_cfg maskrcnn_one_stream configs/mask_rcnn/mask_rcnn_r101_fpn.py 32 --paste-masks --streams 1
mv $OUT/yolov8l_one_stream_conv_layers.json $OUT/yolov8l_conv_layers.json
mv $OUT/maskrcnn_one_stream_conv_layers.json $OUT/maskrcnn_conv_layers.json
python bench.py --config configs/yolov8/yolov8l.py --batch 32 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/yolov8l_bench.json 2> $OUT/yolov8l_bench.err
python bench.py --config configs/mask_rcnn/mask_rcnn_r101_fpn.py --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --paste-masks > $OUT/maskrcnn_bench.json 2> $OUT/maskrcnn_bench.err
tail -c 300 $OUT/yolov8l_bench.json; echo; tail -c 300 $OUT/maskrcnn_bench.json; echo
# (kernel stats on ONE stream: they are what the roofline block's per-launch figures must agree with; a two-stream config's roofline pass is the one-stream pass)
for c in "frcnn configs/faster_rcnn/faster_rcnn_r50_fpn.py 120" "yolov5s configs/yolov5/yolov5s.py 32" "yolov8l configs/yolov8/yolov8l.py 32"; do
  set -- $c
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_$1 -- python3 $ROOT/bench.py --config $ROOT/$2 --batch $3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-zero-operands --no-from-uint8 --streams 1 > $ROOT/$OUT/prof_$1.log 2>&1)
  f=$(ls $OUT/prof_$1/*/*kernel_stats.csv | head -1)
  cp $f $OUT/$1_kernel_stats.csv
  rm -rf $OUT/prof_$1
  head -8 $OUT/$1_kernel_stats.csv | cut -c1-200
done

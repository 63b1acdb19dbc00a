#!/bin/bash
# round-4 evidence, part B (one gpurun call <= 20 min): the secondary configs -- bench lines as shipped, per-layer tables + kernel stats on one stream
set -o pipefail
OUT=gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run_cfg() {  # name config batch [extra bench args]
  n=$1; c=$2; b=$3; shift 3
  python bench.py --config $c --batch $b --steps 10 --warmup 3 --no-cpu-baseline --bracket all --dump-convs $OUT/${n}_conv_layers.json "$@" > $OUT/${n}_all_bench.json 2> $OUT/${n}_all_bench.err
  tail -c 300 $OUT/${n}_all_bench.json; echo
}
run_cfg yolov5s configs/yolov5/yolov5s.py 32
run_cfg yolov8l configs/yolov8/yolov8l.py 32 --streams 1
run_cfg maskrcnn configs/mask_rcnn/mask_rcnn_r101_fpn.py 32 --paste-masks --streams 1
python bench.py --config configs/yolov5/yolov5s.py --batch 32 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/yolov5s_bench.json 2> $OUT/yolov5s_bench.err
python bench.py --config configs/yolov8/yolov8l.py --batch 32 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/yolov8l_bench.json 2> $OUT/yolov8l_bench.err
python bench.py --config configs/mask_rcnn/mask_rcnn_r101_fpn.py --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --paste-masks > $OUT/maskrcnn_bench.json 2> $OUT/maskrcnn_bench.err
for f in yolov5s yolov8l maskrcnn; do python3 -c "
import json; j=json.loads(open('$OUT/${f}_bench.json').read().strip().split('\n')[-1]); print('$f', j['value'], j['ms_per_step'], j['config'].get('streams'), j['roofline']['frac'] if j.get('roofline') else None)"; done
for c in "yolov5s configs/yolov5/yolov5s.py 32" "yolov8l configs/yolov8/yolov8l.py 32"; do
  set -- $c
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_$1 -- python3 $ROOT/bench.py --config $ROOT/$2 --batch $3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-zero-operands --no-from-uint8 --streams 1 > $ROOT/$OUT/prof_$1.log 2>&1)
  f=$(ls $OUT/prof_$1/*/*kernel_stats.csv | head -1); cp $f $OUT/$1_kernel_stats.csv; rm -rf $OUT/prof_$1
  head -4 $OUT/$1_kernel_stats.csv | cut -c1-160
done

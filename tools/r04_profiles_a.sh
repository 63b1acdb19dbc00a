#!/bin/bash
# round-4 evidence, part A (one gpurun call <= 20 min): PMC traffic + default bench line + per-layer table of Faster R-CNN
set -o pipefail
OUT=gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/pmc_conv_traffic.sh $OUT/frcnn_conv_traffic.json 120 --streams 1 && cp $OUT/frcnn_conv_traffic.json profiles/r04_conv_traffic.json
python bench.py --steps 20 --warmup 5 > $OUT/frcnn_bench.json 2> $OUT/frcnn_bench.err
tail -c 600 $OUT/frcnn_bench.json; echo
python bench.py --config configs/faster_rcnn/faster_rcnn_r50_fpn.py --batch 120 --steps 10 --warmup 3 --no-cpu-baseline --bracket all --streams 1 --dump-convs $OUT/frcnn_conv_layers.json > $OUT/frcnn_all_bench.json 2> $OUT/frcnn_all_bench.err
ROOT=$(pwd)
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_frcnn -- python3 $ROOT/bench.py --batch 120 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-zero-operands --no-from-uint8 --streams 1 > $ROOT/$OUT/prof_frcnn.log 2>&1)
f=$(ls $OUT/prof_frcnn/*/*kernel_stats.csv | head -1); cp $f $OUT/frcnn_kernel_stats.csv; rm -rf $OUT/prof_frcnn
head -6 $OUT/frcnn_kernel_stats.csv | cut -c1-200

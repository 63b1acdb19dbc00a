#!/bin/bash
# round-4 evidence, part C: PMC traffic files of the secondary configs (the bench lines of part B read them: run C before B)
set -o pipefail
OUT=gpurun_out/r04
mkdir -p $OUT
bash tools/pmc_conv_traffic.sh $OUT/yolov5s_conv_traffic.json 32 --config configs/yolov5/yolov5s.py && cp $OUT/yolov5s_conv_traffic.json profiles/r04_yolov5s_conv_traffic.json
bash tools/pmc_conv_traffic.sh $OUT/yolov8l_conv_traffic.json 32 --config configs/yolov8/yolov8l.py --streams 1 && cp $OUT/yolov8l_conv_traffic.json profiles/r04_yolov8l_conv_traffic.json
bash tools/pmc_conv_traffic.sh $OUT/maskrcnn_conv_traffic.json 32 --config configs/mask_rcnn/mask_rcnn_r101_fpn.py --streams 1 && cp $OUT/maskrcnn_conv_traffic.json profiles/r04_maskrcnn_conv_traffic.json

#!/bin/bash
# Mask R-CNN R101-FPN: images/s against the per-GPU batch (two streams = the config's default; masks pasted inside the step).
# Why: at b32 the halves are 16 images -> 16 x 4200 / 256 = 262.5 ping-pong tiles on the 23 stage-3 3x3 layers = 1.03 rounds of the 256 CUs.
OUT=${1:-gpurun_out/r04}
mkdir -p $OUT
for b in ${BATCHES:-30 32 60 64 120}; do
  python bench.py --config configs/mask_rcnn/mask_rcnn_r101_fpn.py --batch $b --steps ${STEPS:-8} --warmup 3 --no-cpu-baseline --no-roofline --no-from-uint8 --paste-masks 2> $OUT/mask_sweep_b$b.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask_rcnn_r101 b%d streams %s: %.1f images/s, %.2f ms/step' % ($b, d['config'].get('streams'), d['value'], d['ms_per_step']), flush=True)" | tee -a $OUT/maskrcnn_batch_sweep.txt || exit 1
done

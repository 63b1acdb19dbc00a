# same-box A/B of whole bench.py runs under two values of one environment variable, interleaved.
# usage: bash tools/ab_variant_step.sh <VAR> <value A> <value B> [rounds] [batch] [config]
VAR=$1; A=$2; B=$3; R=${4:-3}; BATCH=${5:-120}; CFG=${6:-configs/faster_rcnn/faster_rcnn_r50_fpn.py}
for r in $(seq $R); do
  for v in $A $B; do
    env $VAR=$v timeout -k 10 300 python bench.py --config $CFG --batch $BATCH --steps 10 --no-cpu-baseline --no-roofline --no-from-uint8 2>/dev/null \
      | grep -o "\"value\": [0-9.]*, \"unit\": \"images/sec\", \"n_gpus\": 1, \"steps\": 10, \"warmup\": 3, \"ms_per_step\": [0-9.]*" | sed "s|^|b$BATCH $VAR=$v: |"
  done
done

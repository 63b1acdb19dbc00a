# same-box comparison of this tree against the round-2 tree checked out at _r02/ (git worktree add -f _r02 f3f56a4 + make there), interleaved.
# usage: bash tools/ab_rounds.sh [rounds]
R=${1:-2}
run() {  # dir config batch
  (cd $1 && timeout -k 10 300 python bench.py --config $2 --batch $3 --steps 10 --no-cpu-baseline --no-roofline --no-from-uint8 2>/dev/null \
    | grep -o "\"value\": [0-9.]*, \"unit\": \"images/sec\", \"n_gpus\": 1, \"steps\": 10, \"warmup\": 3, \"ms_per_step\": [0-9.]*" | sed "s|^|$1 $(basename $2) b$3: |")
}
for r in $(seq $R); do
  for cfg in "configs/faster_rcnn/faster_rcnn_r50_fpn.py 120" "configs/yolov5/yolov5s.py 32" "configs/yolov8/yolov8l.py 32" "configs/mask_rcnn/mask_rcnn_r101_fpn.py 32"; do
    set -- $cfg
    run ${TREE_A:-_r02} $1 $2
    run . $1 $2
  done
done

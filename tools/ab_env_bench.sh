# same-box A/B of whole bench.py runs under an environment switch, interleaved.
# usage: bash tools/ab_env_bench.sh <VAR> <config> <batch> [rounds]      (VAR=1 vs VAR=0)
VAR=$1; CFG=$2; B=$3; R=${4:-2}
for r in $(seq $R); do
  for v in 1 0; do
    env $VAR=$v timeout -k 10 300 python bench.py --config $CFG --batch $B --steps 10 --no-cpu-baseline --no-roofline --no-from-uint8 2>/dev/null \
      | grep -o "\"value\": [0-9.]*, \"unit\": \"images/sec\", \"n_gpus\": 1, \"steps\": 10, \"warmup\": 3, \"ms_per_step\": [0-9.]*" | sed "s|^|$CFG b$B $VAR=$v: |"
  done
done

# same-box A/B of whole bench.py runs: auto dispatch (MD_CONV_VARIANT=0) vs auto without conv1x1_stream_kernel (31), interleaved.
# usage: bash tools/ab_variant_bench.sh <config> <batch> [rounds]
CFG=$1; B=$2; R=${3:-2}
for r in $(seq $R); do
  for v in 0 31; do
    MD_CONV_VARIANT=$v timeout -k 10 300 python bench.py --config $CFG --batch $B --steps 10 --no-cpu-baseline --no-roofline --no-from-uint8 2>/dev/null \
      | grep -o "\"value\": [0-9.]*, \"unit\": \"images/sec\", \"n_gpus\": 1, \"steps\": 10, \"warmup\": 3, \"ms_per_step\": [0-9.]*" | sed "s|^|$CFG b$B variant $v: |"
  done
done

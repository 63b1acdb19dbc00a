# rocprofv3 kernel stats of one bench.py config: bash tools/prof_cfg.sh <name> <config> <batch>  -> gpurun_out/<name>_kernel_stats.csv
export TMPDIR=/tmp
ROOT=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$1 -- python3 $ROOT/bench.py --config $ROOT/$2 --batch $3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-zero-operands --no-from-uint8 > $ROOT/gpurun_out/prof_$1.log 2>&1)
cp $(ls gpurun_out/prof_$1/*/*kernel_stats.csv | head -1) gpurun_out/$1_kernel_stats.csv
rm -rf gpurun_out/prof_$1

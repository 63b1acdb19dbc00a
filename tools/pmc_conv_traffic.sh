#!/bin/bash
# PMC traffic of every conv launch of a bench.py config: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; the guide's
# HBM section: they cannot share a pass, and no tracing domain next to --pmc) -> <out>.json via tools/pmc_traffic.py.
# usage (repo root, GPU box): bash tools/pmc_conv_traffic.sh <out.json> <batch> [bench.py args...]
set -e -o pipefail
OUT=$1; BATCH=$2; shift 2
export TMPDIR=/tmp
ROOT=$(pwd)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $ROOT/gpurun_out/pmc_$C
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_$C -- python3 $ROOT/bench.py --batch $BATCH --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-zero-operands --no-from-uint8 "$@" > $ROOT/gpurun_out/pmc_$C.log 2>&1)
done
python3 tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE 3 $BATCH $OUT > /dev/null
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
python3 -c "
import json; j=json.load(open('$OUT')); print('launches', j['launches'], 'MB per launch', j['hbm_bytes_per_launch']/1e6)
for k,v in j['by_kernel'].items(): print('  %-60s x%4d  %9.1f MB/launch' % (k[:60], v['launches'], v['hbm_bytes_per_launch']/1e6))"

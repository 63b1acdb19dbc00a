#!/bin/bash
# per-kernel PMC averages of one command (separate rocprofv3 passes as the guide prescribes: FETCH_SIZE and WRITE_SIZE cannot share
# a pass; no tracing domains next to --pmc).  usage: bash tools/pmc_kernel.sh <out_prefix> <kernel-substring> -- python3 script args...
set -e -o pipefail
OUT=$1; KSUB=$2; shift 3
export TMPDIR=/tmp
ROOT=$(pwd)
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  TAG=$(echo $C | tr ' ' '_')
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/${OUT}_$TAG -- "$@" > $ROOT/${OUT}_$TAG.log 2>&1) || { tail -5 ${OUT}_$TAG.log; continue; }
  python3 - "$ROOT/${OUT}_$TAG" "$KSUB" <<'PY'
import csv, glob, sys, collections
d, ksub = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*_counter_collection.csv")
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    if ksub in r["Kernel_Name"]:
        k = (r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for k, v in sorted(agg.items()):
    print("PMC", k[0], k[1], "avg per launch", v[0] / v[1], "launches", v[1])
PY
  rm -rf ${OUT}_$TAG
done

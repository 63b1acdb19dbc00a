#!/bin/bash
# several PMC passes (one counter group each) of one command, per-kernel averages.  usage: bash tools/pmc_multi.sh <kernel-substring> <script + args>
# groups are separate rocprofv3 runs (--pmc with --kernel-trace only, as the pool requires)
set -o pipefail
KSUB=$1; shift
export TMPDIR=/tmp
ROOT=$(pwd)
# (eight SQ counters in one pass aborted rocprofv3 on this pool: four at most per pass, each pass under its own timeout)
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE"; do
  D=$ROOT/gpurun_out/pmcm_$$
  rm -rf $D
  (cd /tmp && timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $ROOT/"$@" > $D.log 2>&1) || { echo "pass failed: $C"; tail -3 $D.log | cut -c1-300; rm -rf $D $D.log; continue; }
  python3 - "$D" "$KSUB" <<'PY'
import csv, glob, sys, collections
d, ksub = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*_counter_collection.csv")
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    if ksub in r["Kernel_Name"]:
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for k, v in sorted(agg.items()):
    print("PMC %-42s %-40s %16.0f per launch (%d launches)" % (k[0], k[1], v[0] / v[1], v[1]))
PY
  rm -rf $D $D.log
done

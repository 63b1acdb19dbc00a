#!/bin/bash
# whole-step survey of LDS bank conflicts per kernel: one rocprofv3 PMC pass (--pmc with --kernel-trace only) over a short bench.py run,
# aggregated by kernel name.  conflict share = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS (cycles lost to conflicts per LDS-busy cycle).
# usage: bash tools/pmc_lds_survey.sh [bench.py args...]
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
D=$ROOT/gpurun_out/pmclds_$$
rm -rf $D
(cd /tmp && timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $D -- \
   python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-from-uint8 "$@" > $D.log 2>&1) || { echo "pass failed"; tail -5 $D.log | cut -c1-300; }
python3 - "$D" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][-70:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        n[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])
print("%-72s %7s %14s %14s %9s %9s" % ("kernel", "launch", "wave_cycles", "bank_conflict", "conf/lds", "lds/wave"))
for k, v in rows[:40]:
    a = v["SQ_ACTIVE_INST_LDS"]
    print("%-72s %7d %14.0f %14.0f %9.3f %9.3f" % (k, n[k], v["SQ_WAVE_CYCLES"], v["SQ_LDS_BANK_CONFLICT"], v["SQ_LDS_BANK_CONFLICT"] / a if a else 0.0,
                                                     a / v["SQ_WAVE_CYCLES"] if v["SQ_WAVE_CYCLES"] else 0.0))
PY
rm -rf $D $D.log

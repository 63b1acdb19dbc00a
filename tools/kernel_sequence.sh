#!/bin/bash
# kernel launch sequence of ONE bench step (rocprofv3 kernel trace, sorted by start time): which launches sit between the convs
# usage: bash tools/kernel_sequence.sh <config> <batch> <out.txt>
CFG=$1; B=$2; OUT=$3; shift 3   # further arguments go to bench.py (e.g. --streams 1)
ROOT=$(pwd); D=$ROOT/gpurun_out/kseq; rm -rf $D; mkdir -p $D $(dirname $OUT)
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $ROOT/bench.py --config $ROOT/$CFG --batch $B --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-zero-operands --no-from-uint8 "$@" > $D/log.txt 2>&1)
f=$(ls $D/*/*kernel_trace.csv | head -1)
python3 - "$f" "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last step = from the last stem kernel on
idx = max(i for i, n in enumerate(names) if "stem" in n)
out = open(sys.argv[2], "w")
t0 = int(rows[idx]["Start_Timestamp"])
prev_end = t0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.write(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'][:110]}\n")
    prev_end = e
print("kernels in the last step:", len(rows) - idx, " span", (prev_end - t0) / 1e3, "us")
PY
rm -rf $D

# rocprofv3 kernel stats of one config in this tree and in _r02/, same box.  usage: bash tools/prof_two_trees.sh <config> <batch>
export TMPDIR=/tmp
ROOT=$(pwd)
for T in _r02 .; do
  D=$ROOT/gpurun_out/ptt_$(echo $T | tr -d './')x
  rm -rf $D
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $ROOT/$T/bench.py --config $ROOT/$T/$1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-zero-operands --no-from-uint8 > $D.log 2>&1)
  f=$(ls $D/*/*kernel_stats.csv | head -1)
  echo "== tree $T: $(grep -o '"ms_per_step": [0-9.]*' $D.log | head -1)"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("   total kernel time per step (7 steps profiled): %.3f ms" % (tot / 7e6))
for r in rows[:22]:
    print("   %-90s x%4s %9.3f ms/step" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 7e6))
PY
  rm -rf $D
done

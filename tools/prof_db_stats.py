"""Per-kernel totals from rocprofv3's rocpd sqlite output (kernels view): python tools/prof_db_stats.py results.db [results2.db] [--steps N]
With two databases prints both and the difference (A/B of two runs of the same command)."""
import sqlite3, sys, collections, re

def load(path):
    cur = sqlite3.connect(path).cursor()
    d = collections.defaultdict(lambda: [0, 0])
    for name, grid, dur in cur.execute("select name, grid_x, duration from kernels"):
        name = re.sub(r"\(.*", "", name)
        d[name][0] += 1; d[name][1] += dur
    return d

args = [a for a in sys.argv[1:] if a.endswith(".db")]
steps = float(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1.0
dbs = [load(a) for a in args]
names = sorted(set().union(*[set(d) for d in dbs]), key=lambda n: -max(d[n][1] for d in dbs))
tot = [sum(v[1] for v in d.values()) for d in dbs]
print("total ms/step: " + "  ".join(f"{t/steps/1e6:.3f}" for t in tot))
for n in names[:40]:
    line = f"{n[:70]:70s}"
    for d in dbs:
        c, t = d[n]
        line += f"  {c/steps:6.1f}x {t/steps/1e6:7.3f} ms"
    if len(dbs) == 2:
        line += f"  delta {(dbs[1][n][1]-dbs[0][n][1])/steps/1e6:+.3f}"
    print(line)

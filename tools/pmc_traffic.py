"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into the
HBM traffic of the dominant kernel, corrected as MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
FETCH_SIZE (KB) reports half of the bytes of wide coalesced reads -> x2; WRITE_SIZE (KB) is exact for 16-B
streaming stores.  Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <steps_profiled> <batch_per_gpu> <out.json>"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][0] += float(r["Counter_Value"])
        agg[r["Kernel_Name"]][1] += 1
    return agg


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
steps = int(sys.argv[3])
batch = int(sys.argv[4])
conv = [k for k in fetch if "conv_igemm_kernel" in k or "conv3x3_halo_kernel" in k or "conv_pingpong_kernel" in k or "bottleneck64_kernel" in k or "conv1x1_stream_kernel" in k or "c3pair" in k]
f_kb = sum(fetch[k][0] for k in conv)
w_kb = sum(write[k][0] for k in conv)
n = sum(fetch[k][1] for k in conv)
out = {"kernel": "conv_pingpong_kernel + conv_igemm_kernel + conv3x3_halo_kernel + bottleneck64_kernel + conv1x1_stream_kernel + c3pair kernels (all conv/FC launches)", "profiled_steps": steps,
       "batch_per_gpu": batch, "launches": n,
       "FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB": w_kb, "fetch_correction": 2.0,
       "hbm_bytes_per_step": (2.0 * f_kb + w_kb) * 1024 / steps,
       "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024 / n}
by = {}
for k in conv:
    name = k.split("(")[0].replace("void md::", "").strip()  # keeps the template arguments: one entry per instantiation
    d = by.setdefault(name, [0.0, 0.0, 0])
    d[0] += fetch[k][0]; d[1] += write[k][0]; d[2] += fetch[k][1]
out["by_kernel"] = {n_: {"launches": d[2], "FETCH_SIZE_KB_raw": d[0], "WRITE_SIZE_KB": d[1],
                         "hbm_bytes_per_launch": (2.0 * d[0] + d[1]) * 1024 / d[2]} for n_, d in by.items()}
# the build these counters describe: bench.py reports traffic only when the kernel sources it runs hash to the same value
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out["kernel_source_sha256"] = bench.kernel_source_hash()
json.dump(out, open(sys.argv[5], "w"), indent=1)
print(out)

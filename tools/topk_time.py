"""md_topk_segmented on YOLO-shaped segments: time per launch (MD_LIB_OVERRIDE = another build of the library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import det_ops, _lib
if os.environ.get("MD_LIB_OVERRIDE"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MD_LIB_OVERRIDE"])
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for B, n, k, thr in ((32, 25200, 4096, 0.25), (32, 8400, 4096, 0.25), (16, 8400, 1024, 0.05), (32, 25200, 4096, 0.9)):
    sc = torch.rand((B * n,), generator=g).to(dev)
    seg = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev)
    fn = lambda: det_ops.topk_segmented(sc, seg, k, min_score=thr, max_segment=n)
    v, i, c = fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{B} segments x {n} scores, k {k}, min_score {thr}: selected {int(c[0])} -> {sorted(ts)[2]:7.1f} us", flush=True)

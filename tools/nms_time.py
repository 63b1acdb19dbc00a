"""Greedy-NMS launch times on the shapes the detectors use: batched class-wise lists with a quota (one-stage / RPN), and the reference's rotated NMS
(NmsGpu signature) on long lists without a quota."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import det_ops, _lib
if os.environ.get("MD_LIB_OVERRIDE"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MD_LIB_OVERRIDE"])
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
ev = lambda: torch.cuda.Event(enable_timing=True)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[2]


def boxes(B, n, span):
    c = torch.rand((B, n, 2), generator=g) * span
    wh = torch.rand((B, n, 2), generator=g) * 60 + 8
    return torch.cat([c - wh / 2, c + wh / 2], 2).to(dev)


for B, n, quota, span in ((32, 4096, 300, 640.0), (120, 4750, 1000, 1344.0), (120, 1000, 100, 1344.0), (1, 30000, 0, 4000.0)):
    b = boxes(B, n, span)
    grp = torch.randint(0, 80, (B, n), generator=g, dtype=torch.int32).to(dev) if quota == 300 else None
    cnt = torch.full((B,), n, dtype=torch.int32, device=dev)
    fn = lambda: det_ops.nms_aligned(b, 0.5, mode=det_ops.NMS_MODE_STRICT, count=cnt, group=grp, max_output=quota)
    m, i, num = fn()
    print(f"{B} lists x {n} boxes, quota {quota}: kept {int(num[0])} -> {timed(fn):8.1f} us per call (mask + scan launches)", flush=True)

"""The second half of a stage-3 bottleneck block (3x3 256->256 + ReLU -> 1x1 256->1024 + residual + ReLU): the two launches, each alone,
against md_conv2d_expand (one launch), with a cache-flushing fill before every timed call.  python tools/expand_ab.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
H, W = 50, 84
pc2 = nn_ops.pack_conv(torch.randn((256, 256, 3, 3), generator=g) * (2.0 / 2304) ** 0.5, stride=1, pad=1, relu=True).to(dev)
pc3 = nn_ops.pack_conv(torch.randn((1024, 256, 1, 1), generator=g) * (2.0 / 256) ** 0.5, relu=True).to(dev)
x = torch.relu(torch.randn((B, H, W, 256), generator=g)).to(torch.bfloat16).to(dev)
r = torch.relu(torch.randn((B, H, W, 1024), generator=g)).to(torch.bfloat16).to(dev)
scratch = torch.empty((1536 << 20,), dtype=torch.uint8, device=dev)
t2 = nn_ops.conv2d(x, pc2)
ref = nn_ops.conv2d(t2, pc3, residual=r)
assert torch.equal(nn_ops.conv2d_expand(x, pc2, pc3, residual=r), ref)
arms = {"conv2 3x3": lambda: nn_ops.conv2d(x, pc2), "conv3 1x1+res": lambda: nn_ops.conv2d(t2, pc3, residual=r),
        "pair (two launches)": lambda: nn_ops.conv2d_expand(x, pc2, pc3, residual=r, variant=31),
        "fused expand": lambda: nn_ops.conv2d_expand(x, pc2, pc3, residual=r)}
lib = _lib.lib()
def staggered(u):
    def f():
        lib.md_conv2d_set_expand_stagger(u)
        nn_ops.conv2d_expand(x, pc2, pc3, residual=r)
        lib.md_conv2d_set_expand_stagger(2)
    return f
for u in (0, 1, 3, 4, 6):
    arms[f"fused, stagger {u}"] = staggered(u)
times = {k: [] for k in arms}
for rnd in range(7):
    for k, f in arms.items():
        scratch.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1))
for k in arms:
    print(f"{k:22s} {sorted(times[k])[3]*1e3:8.1f} us", flush=True)

"""Interleaved A/B of md_conv2d variants on one layer shape (N(0,1) data, optional residual), with a cache flush between launches so that the
activation tensor does not survive in the Infinity Cache (the in-situ condition, DESIGN_HISTORY 6b).
python tools/layer_ab.py N H W Cin Cout k res(0|1) v1,v2,... [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops, _lib
N, H, W, Cin, Cout, k, res = [int(v) for v in sys.argv[1:8]]
VARS = [int(v) for v in sys.argv[8].split(",")]
ROUNDS = int(sys.argv[9]) if len(sys.argv) > 9 else 7
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (k * k * Cin)) ** 0.5
pc = nn_ops.pack_conv(w, bias=torch.randn((Cout,), generator=g) * 0.1, stride=1, pad=k // 2, relu=True).to(dev)
x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
r = torch.randn((N, H, W, Cout), generator=g).to(torch.bfloat16).to(dev) if res else None
flush = torch.empty((1536 << 20,), dtype=torch.uint8, device=dev)
ref = nn_ops.conv2d(x, pc, residual=r, variant=VARS[0]).clone()
y = torch.empty_like(ref)
t = {v: [] for v in VARS}
same, kern = {}, {}
for rd in range(ROUNDS):
    for v in VARS:
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nn_ops.conv2d(x, pc, residual=r, variant=v, out=y)
        e1.record(); torch.cuda.synchronize()
        kern[v] = _lib.lib().md_conv2d_last_kernel()
        t[v].append(e0.elapsed_time(e1)); same[v] = torch.equal(y, ref)
fl = 2.0 * ref.numel() * Cin * k * k
by = 2.0 * (x.numel() + ref.numel() * (2 if res else 1))
for v in VARS:
    s = sorted(t[v]); m = s[len(s) // 2]
    print(f"{N}x{H}x{W} {Cin}->{Cout} k{k} res{res} | variant {v:2d} (kernel id {kern[v]}): median {m*1e3:8.1f} us  min {s[0]*1e3:8.1f}  {fl/m/1e9:7.1f} TF  {by/m/1e9:6.2f} TB/s  identical to first: {same[v]}", flush=True)

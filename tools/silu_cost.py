"""What the SiLU epilogue costs: md_conv2d on YOLO-shaped layers with act = SiLU against the same layer with ReLU / no activation (same kernel, same traffic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
ev = lambda: torch.cuda.Event(enable_timing=True)
for (N, H, W, Cin, Cout, k, s) in [(32, 160, 160, 64, 64, 1, 1), (32, 320, 320, 32, 64, 3, 2), (32, 160, 160, 64, 128, 3, 2), (32, 80, 80, 128, 128, 1, 1), (32, 80, 80, 128, 128, 3, 1),
                                   (32, 40, 40, 256, 256, 1, 1), (32, 40, 40, 256, 256, 3, 1), (32, 80, 80, 256, 256, 3, 1)]:
    w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
    line = f"{N}x{H}x{W} {Cin}->{Cout} k{k} s{s}:"
    for act in ("silu", True, False):
        pc = nn_ops.pack_conv(w, bias=torch.zeros(Cout), stride=s, pad=k // 2, relu=act).to(dev)
        y = nn_ops.conv2d(x, pc)
        ts = []
        for _ in range(7):
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(10):
                nn_ops.conv2d(x, pc, out=y)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        line += f"  {act if act else 'none'}: {sorted(ts)[3]:6.1f} us"
    print(line, flush=True)

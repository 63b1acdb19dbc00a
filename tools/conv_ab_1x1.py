"""A/B on the residual 1x1 expand / reduce layers: auto (0) vs the ping-pong kernel forced (15) vs two staging buffers (2).
python tools/conv_ab_1x1.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
LAYERS = [(100, 168, 128, 512, True), (50, 84, 256, 1024, True), (25, 42, 512, 2048, True), (50, 84, 1024, 256, False), (25, 42, 2048, 512, False),
          (100, 168, 512, 256, False), (50, 84, 1024, 512, False), (25, 42, 1024, 2048, False), (50, 84, 512, 1024, False)]
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for (H, W, Cin, Cout, res) in LAYERS:
    w = torch.randn((Cout, Cin, 1, 1), generator=g) * (2.0 / Cin) ** 0.5
    pc = nn_ops.pack_conv(w, relu=True).to(dev)
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
    r = torch.randn((B, H, W, Cout), generator=g).to(torch.bfloat16).to(dev) if res else None
    by = 2.0 * B * H * W * (Cin + Cout * (2 if res else 1))
    variants = [0, 2, 15]
    times = {v: [] for v in variants}
    for v in variants:
        nn_ops.conv2d(x, pc, residual=r, variant=v)
    for rnd in range(5):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                nn_ops.conv2d(x, pc, residual=r, variant=v)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 4)
    line = f"{B}x{H}x{W}x{Cin}->{Cout} k1{' +res' if res else ''}:"
    for v in variants:
        t = sorted(times[v])[2]
        line += f"  v{v} {t*1e3:7.1f}us {by/t/1e9:6.2f}TB/s"
    print(line, flush=True)

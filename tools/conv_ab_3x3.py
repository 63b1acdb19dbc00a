"""A/B of kernel variants on the Faster R-CNN / Mask R-CNN 3x3 layers (korder-1 weights, ReLU): auto (0) against the halo kernel with
64-cout tiles (27), ping-pong (15), 128x128 with one / two staging buffers (20 / 2).  python tools/conv_ab_3x3.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
LAYERS = [(200, 336, 256, 256), (100, 168, 256, 256), (50, 84, 256, 256), (25, 42, 256, 256), (13, 21, 256, 256), (100, 168, 128, 128),
          (50, 84, 256, 256), (25, 42, 512, 512), (200, 336, 64, 64)]
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for (H, W, Cin, Cout) in LAYERS:
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (9 * Cin)) ** 0.5
    pc = nn_ops.pack_conv(w, stride=1, pad=1, relu=True).to(dev)
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
    fl = 2.0 * B * H * W * Cout * Cin * 9
    variants = [0, 20, 2, 27] + ([15] if Cout % 256 == 0 else [])
    times = {v: [] for v in variants}
    for v in variants:
        nn_ops.conv2d(x, pc, variant=v)
    for rnd in range(5):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                nn_ops.conv2d(x, pc, variant=v)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 3)
    line = f"{B}x{H}x{W}x{Cin}->{Cout} k3:"
    for v in variants:
        t = sorted(times[v])[2]
        line += f"  v{v} {t*1e3:7.1f}us {fl/t/1e9:6.0f}TF"
    print(line, flush=True)

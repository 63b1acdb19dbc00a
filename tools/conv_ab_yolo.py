"""A/B of kernel variants on the YOLOv8l 3x3 SiLU layers (korder-1 weights): auto (0), halo kernel with 128- / 64-cout tiles (11 / 27),
ping-pong (15), 128x128 with two staging buffers (2).  python tools/conv_ab_yolo.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
LAYERS = [(80, 80, 128, 128), (40, 40, 256, 256), (20, 20, 256, 256), (160, 160, 64, 64), (80, 80, 256, 256), (40, 40, 512, 512), (20, 20, 512, 512)]
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for (H, W, Cin, Cout) in LAYERS:
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (9 * Cin)) ** 0.5
    pc = nn_ops.pack_conv(w, stride=1, pad=1, relu="silu").to(dev)
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
    fl = 2.0 * B * H * W * Cout * Cin * 9
    variants = [0, 2, 11, 27] + ([15] if Cout % 256 == 0 else [])
    ref = nn_ops.conv2d(x, pc, variant=20)
    times = {v: [] for v in variants}
    for v in variants:
        y = nn_ops.conv2d(x, pc, variant=v)
        assert (y.float() - ref.float()).abs().max() <= 0.05 * ref.float().abs().max(), v
    for rnd in range(5):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                nn_ops.conv2d(x, pc, variant=v)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5)
    line = f"{B}x{H}x{W}x{Cin}->{Cout} k3 silu:"
    for v in variants:
        t = sorted(times[v])[2]
        line += f"  v{v} {t*1e3:7.1f}us {fl/t/1e9:6.0f}TF"
    print(line, flush=True)

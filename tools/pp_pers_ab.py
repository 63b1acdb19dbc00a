"""A/B of the persistent form of the ping-pong kernel (variant 32) against the dispatcher's choice without it (variant 33), interleaved,
with a correctness check (bit-identical).  python tools/pp_pers_ab.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
LAYERS = [(200, 336, 256, 256, 3, 30), (100, 168, 256, 256, 3, B), (50, 84, 256, 256, 3, B), (50, 84, 1024, 256, 1, B), (25, 42, 512, 512, 3, B),
          (25, 42, 2048, 512, 1, B), (50, 84, 1024, 512, 1, B)]
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for (H, W, Cin, Cout, k, b) in LAYERS:
    w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    pc = nn_ops.pack_conv(w, bias=torch.randn((Cout,), generator=g) * 0.1, stride=1, pad=k // 2, relu=True).to(dev)
    x = torch.relu(torch.randn((b, H, W, Cin), generator=g)).to(torch.bfloat16).to(dev)
    y = torch.empty((b, H, W, Cout), dtype=torch.bfloat16, device=dev)
    ref = nn_ops.conv2d(x, pc, variant=33).clone()
    got = nn_ops.conv2d(x, pc, variant=32)
    torch.cuda.synchronize()
    same = torch.equal(ref, got)
    fl = 2.0 * b * H * W * Cout * Cin * k * k
    times = {33: [], 32: []}
    for rnd in range(5):
        for v in (33, 32):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                nn_ops.conv2d(x, pc, variant=v, out=y)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 4)
    t0, t1 = sorted(times[33])[2], sorted(times[32])[2]
    print(f"{b}x{H}x{W}x{Cin}->{Cout} k{k}: one tile per WG {t0*1e3:8.1f} us ({fl/t0/1e9:6.0f} TF)  persistent {t1*1e3:8.1f} us ({fl/t1/1e9:6.0f} TF)  ratio {t0/t1:.3f}  identical={same}", flush=True)

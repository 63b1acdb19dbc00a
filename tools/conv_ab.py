"""A/B of the conv kernel variants on the benchmark's layer shapes, interleaved rounds in ONE process
(cdna_hip_programming.md rule 24).  Usage: python tools/conv_ab.py [batch]"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from minddet_amd import nn_ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
LAYERS = [  # (H, W, Cin, Cout, k, stride, residual)
    (200, 336, 256, 256, 3, 1, False), (100, 168, 256, 256, 3, 1, False), (50, 84, 256, 256, 3, 1, False),
    (200, 336, 64, 256, 1, 1, True), (200, 336, 64, 256, 1, 1, False), (200, 336, 64, 64, 3, 1, False),
    (200, 336, 256, 64, 1, 1, False), (100, 168, 128, 512, 1, 1, True), (100, 168, 128, 128, 3, 1, False),
    (50, 84, 256, 1024, 1, 1, True), (50, 84, 1024, 256, 1, 1, False), (25, 42, 512, 512, 3, 1, False),
    (25, 42, 512, 2048, 1, 1, True), (25, 42, 2048, 512, 1, 1, False), (1, 1000 * B // 8, 12544, 1024, 1, 1, False),
    (1, 1000 * B // 8, 1024, 1024, 1, 1, False), (200, 336, 256, 256, 1, 1, True), (100, 168, 512, 128, 1, 1, False),
]
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
rows = []
for (H, W, Cin, Cout, k, s, res) in LAYERS:
    w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    pc = nn_ops.pack_conv(w, stride=s, pad=k // 2, relu=True, korder=0).to(dev)
    pc1 = nn_ops.pack_conv(w, stride=s, pad=k // 2, relu=True, korder=1).to(dev) if (Cin % 64 == 0 and k > 1) else None
    n = 8 if H == 1 else B
    x = torch.randn((n, H, W, pc.cin), generator=g).to(torch.bfloat16).to(dev)
    ho, wo = nn_ops.conv_out_hw(H, W, pc)
    r = torch.randn((n, ho, wo, pc.cout), generator=g).to(torch.bfloat16).to(dev) if res else None
    fl = 2.0 * n * ho * wo * Cout * Cin * k * k
    variants = [0, 2, 15, 20] if nn_ops.cout_tile(pc.cout) == 128 else [0, 2, 20]
    outs, times = {}, {v: [] for v in variants}
    def run(v):
        if v >= 100:
            return nn_ops.conv2d(x, pc1 if pc1 is not None else pc, residual=r, variant=v - 100)
        return nn_ops.conv2d(x, pc, residual=r, variant=v)
    for v in variants:
        outs[v] = run(v)
    torch.cuda.synchronize()
    for v in []:
        assert torch.equal(outs[v], outs[variants[0]]) or (outs[v].float() - outs[variants[0]].float()).abs().max() < 1e-1, (v, "mismatch")
    for rnd in range(5):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(v)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 3)
    line = f"{n}x{H}x{W}x{Cin}->{Cout} k{k}s{s}{' +res' if res else ''}:"
    for v in variants:
        t = sorted(times[v])[len(times[v]) // 2]
        line += f"  v{v} {t*1e3:8.1f}us {fl/t/1e9:7.1f}TF"
    print(line, flush=True)

"""A/B of the two MFMA shapes in the ping-pong kernel (variant 15 = v_mfma_f32_32x32x16_bf16, 22 = v_mfma_f32_16x16x32_bf16) on the
benchmark's MFMA-bound layers, interleaved rounds in one process, random post-ReLU-like data."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from minddet_amd import nn_ops

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
LAYERS = [(200, 336, 256, 256, 3), (100, 168, 256, 256, 3), (50, 84, 256, 256, 3), (25, 42, 512, 512, 3), (50, 84, 1024, 256, 1),
          (50, 84, 1024, 2048, 1), (25, 42, 2048, 512, 1)]
g = torch.Generator().manual_seed(0)
for (H, W, Cin, Cout, k) in LAYERS:
    w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    pc = nn_ops.pack_conv(w, bias=torch.zeros(Cout), stride=1, pad=k // 2, relu=True, korder=1 if k == 3 else 0).to(dev)
    x = torch.relu(torch.randn((B, H, W, Cin), generator=torch.Generator(device=dev).manual_seed(1), device=dev)).to(torch.bfloat16)
    y = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=dev)
    outs = {}
    t = {15: [], 22: []}
    for v in t:
        outs[v] = nn_ops.conv2d(x, pc, variant=v, out=y).clone()
    for _ in range(6):
        for v in t:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _i in range(4):
                nn_ops.conv2d(x, pc, variant=v, out=y)
            e1.record()
            torch.cuda.synchronize()
            t[v].append(e0.elapsed_time(e1) / 4)
    fl = 2.0 * B * H * W * Cout * Cin * k * k
    m15, m22 = sorted(t[15])[3], sorted(t[22])[3]
    d = (outs[15].float() - outs[22].float()).abs().max().item()
    print(f"{H}x{W} {Cin}->{Cout} k{k}: 32x32x16 {m15:.3f} ms ({fl / m15 / 1e9:.0f} TF)  16x16x32 {m22:.3f} ms ({fl / m22 / 1e9:.0f} TF)  ratio {m15 / m22:.3f}  max diff {d:.3g}")

"""A/B of md_bottleneck (one launch per ResNet stage-1 block) against the three md_conv2d launches it replaces, interleaved rounds in
ONE process (guide 5.4 rule 24), random data.  Usage: python tools/bottleneck_ab.py [batch H W]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from minddet_amd import nn_ops

B, H, W = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (60, 200, 336)
dev = "cuda:0"
g = torch.Generator().manual_seed(0)


def mods(cin, ds):
    w1 = torch.randn((64, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5
    w2 = torch.randn((64, 64, 3, 3), generator=g) * (2.0 / 576) ** 0.5
    w3 = torch.randn((256, 64, 1, 1), generator=g) * (2.0 / 64) ** 0.5
    pcs = [nn_ops.pack_conv(w1, bias=torch.zeros(64), relu=True).to(dev), nn_ops.pack_conv(w2, bias=torch.zeros(64), stride=1, pad=1, relu=True).to(dev),
           nn_ops.pack_conv(w3, bias=torch.zeros(256), relu=True).to(dev)]
    pd = nn_ops.pack_conv(torch.randn((256, cin, 1, 1), generator=g) * (1.0 / cin) ** 0.5, bias=torch.zeros(256), relu=False).to(dev) if ds else None
    return pcs, pd


for cin, ds in ((256, False), (64, True)):
    (pc1, pc2, pc3), pd = mods(cin, ds)
    blk = nn_ops.pack_bottleneck(pc1, pc2, pc3, pd)
    x = torch.relu(torch.randn((B, H, W, cin), generator=torch.Generator(device=dev).manual_seed(1), device=dev)).to(torch.bfloat16)
    y = torch.empty((B, H, W, 256), dtype=torch.bfloat16, device=dev)

    def three():
        res = nn_ops.conv2d(x, pd) if ds else x
        return nn_ops.conv2d(nn_ops.conv2d(nn_ops.conv2d(x, pc1), pc2), pc3, residual=res, out=y)

    def fused():
        return nn_ops.bottleneck(x, blk, out=y)   # the downsample conv (if any) is computed inside the launch

    a = three().clone()
    b = fused().clone()
    torch.cuda.synchronize()
    print(f"Cin {cin} downsample {ds}: identical {torch.equal(a, b)}")
    t = {"three": [], "fused": []}
    for _ in range(5):
        for name, fn in (("three", three), ("fused", fused)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _i in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t[name].append(e0.elapsed_time(e1) / 5)
    px = B * H * W
    for name in t:
        ms = sorted(t[name])[len(t[name]) // 2]
        byts = px * 2 * (cin + 256)   # the fused op's algorithmic bytes: x read once, y written once
        print(f"   {name:6s} median {ms:.3f} ms  min {min(t[name]):.3f} ms   ({byts / ms / 1e9:.2f} TB/s of the fused op's algorithmic bytes)")

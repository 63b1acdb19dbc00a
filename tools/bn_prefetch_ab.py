"""bottleneck64_kernel: L2 warm-up of a later tile's first x chunks (MD_BN_PREFETCH = tiles ahead on the same XCD; diagnostic library only) against none,
interleaved rounds in one process.  MD_DIAG_LIB=1 python tools/bn_prefetch_ab.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MD_DIAG_LIB"] = "1"
import torch
from minddet_amd import nn_ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
H, W, dev = 200, 336, "cuda:0"
g = torch.Generator().manual_seed(0)
PFS = [0, 64, 128, 256, 32]
for cin, ds in ((256, False), (64, True)):
    w1 = torch.randn((64, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5
    w2 = torch.randn((64, 64, 3, 3), generator=g) * (2.0 / 576) ** 0.5
    w3 = torch.randn((256, 64, 1, 1), generator=g) * (2.0 / 64) ** 0.5
    pcs = [nn_ops.pack_conv(w1, bias=torch.zeros(64), relu=True).to(dev), nn_ops.pack_conv(w2, bias=torch.zeros(64), stride=1, pad=1, relu=True).to(dev),
           nn_ops.pack_conv(w3, bias=torch.zeros(256), relu=True).to(dev)]
    pd = nn_ops.pack_conv(torch.randn((256, cin, 1, 1), generator=g) * (1.0 / cin) ** 0.5, bias=torch.zeros(256), relu=False).to(dev) if ds else None
    blk = nn_ops.pack_bottleneck(*pcs, pd)
    x = torch.relu(torch.randn((B, H, W, cin), generator=torch.Generator(device=dev).manual_seed(1), device=dev)).to(torch.bfloat16)
    y = torch.empty((B, H, W, 256), dtype=torch.bfloat16, device=dev)
    os.environ["MD_BN_PREFETCH"] = "0"
    ref = nn_ops.bottleneck(x, blk).clone()
    t = {pf: [] for pf in PFS}
    same = {}
    for r in range(6):
        for pf in PFS:
            os.environ["MD_BN_PREFETCH"] = str(pf)
            nn_ops.bottleneck(x, blk, out=y)
            torch.cuda.synchronize()
            same[pf] = torch.equal(y, ref)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                nn_ops.bottleneck(x, blk, out=y)
            e1.record(); torch.cuda.synchronize()
            t[pf].append(e0.elapsed_time(e1) / 5)
    print(f"Cin {cin} downsample {ds}, batch {B}:", flush=True)
    for pf in PFS:
        v = sorted(t[pf])
        print(f"   prefetch {pf:4d} tiles ahead: median {v[len(v)//2]:.3f} ms  min {v[0]:.3f}  identical {same[pf]}", flush=True)

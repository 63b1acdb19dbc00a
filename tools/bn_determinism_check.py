"""md_bottleneck determinism / parity at the benchmark's stage-1 shape, product or diagnostic library (MD_DIAG_LIB=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops, _lib
if os.environ.get('MD_LIB_OVERRIDE'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MD_LIB_OVERRIDE'])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for cin, ds in ((256, False), (64, True)):
    w1 = torch.randn((64, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5
    w2 = torch.randn((64, 64, 3, 3), generator=g) * (2.0 / 576) ** 0.5
    w3 = torch.randn((256, 64, 1, 1), generator=g) * (2.0 / 64) ** 0.5
    pcs = [nn_ops.pack_conv(w1, bias=torch.zeros(64), relu=True).to(dev), nn_ops.pack_conv(w2, bias=torch.zeros(64), stride=1, pad=1, relu=True).to(dev),
           nn_ops.pack_conv(w3, bias=torch.zeros(256), relu=True).to(dev)]
    pd = nn_ops.pack_conv(torch.randn((256, cin, 1, 1), generator=g) * (1.0 / cin) ** 0.5, bias=torch.zeros(256), relu=False).to(dev) if ds else None
    blk = nn_ops.pack_bottleneck(*pcs, pd)
    x = torch.relu(torch.randn((B, 200, 336, cin), generator=torch.Generator(device=dev).manual_seed(1), device=dev)).to(torch.bfloat16)
    res = nn_ops.conv2d(x, pd) if ds else x
    three = nn_ops.conv2d(nn_ops.conv2d(nn_ops.conv2d(x, pcs[0]), pcs[1]), pcs[2], residual=res)
    a = nn_ops.bottleneck(x, blk).clone()
    y = torch.empty_like(a)
    torch.cuda.synchronize()
    print(f"Cin {cin} ds {ds}: fused == three launches: {torch.equal(a, three)}; nan in fused {bool(torch.isnan(a.float()).any())}; nan in three {bool(torch.isnan(three.float()).any())}")
    for r in range(4):
        y.fill_(3.0)
        nn_ops.bottleneck(x, blk, out=y)
        torch.cuda.synchronize()
        d = (y != a)
        print(f"   run {r}: equal to first run {torch.equal(y, a)}; differing elements {int(d.sum())}")
        if d.any():
            idx = d.nonzero()[:6].tolist()
            print("      first differing:", idx, [ (float(y[tuple(i)]), float(a[tuple(i)])) for i in idx])
        if d.any() and r < 2:
            idx = d.nonzero().cpu()
            import collections
            pl = (idx[:, 1] % 8) * 16 + idx[:, 2] % 16
            hist = collections.Counter()
            for p, c in zip(pl.tolist(), idx[:, 3].tolist()):
                hist[(p // 32, p % 32, c // 64, (c % 64) // 32, ((c % 64) % 32) // 8, c % 8)] += 1
            print("      (wq, pixel in wave, quarter, f, g, channel in group of 8) -> count; tiles:", len({(n_, y_ // 8, x_ // 16) for n_, y_, x_, _ in idx.tolist()}))
            agg = collections.Counter()
            for (wq, pw, q, f, g_, ce), n_ in hist.items():
                agg[(wq, pw, q, f, ce)] += n_
            for k_, v_ in sorted(agg.items(), key=lambda kv: -kv[1])[:80]:
                print("        ", k_, v_)
            tl = collections.Counter((n_, y_ // 8, x_ // 16) for n_, y_, x_, _ in idx.tolist())
            print("      tiles (n, ty, tx) sample:", sorted(tl)[:12], "pt % 64 of bad tiles:", sorted(collections.Counter(((n_ * 25 + ty_) * 21 + tx_) % 8 for (n_, ty_, tx_) in tl).items()))

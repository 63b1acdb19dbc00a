"""Race hunt: the persistent ping-pong form (variant 32), its HALO forms (36 / 37 / 38, the fused head with 34), the halo-reuse kernel (27 / 11), the stream kernel (30), md_bottleneck
and md_stem_conv run many times on the same operands;
every output must equal the first one and the reference kernel's bit for bit.  python tools/stress_new_kernels.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
bad = 0
CASES = [("pers 3x3 256 b30 P2", 32, 15, (30, 200, 336, 256), 256, 3, False), ("pers 3x3 256 @50x84", 32, 15, (60, 50, 84, 256), 256, 3, False),
         ("pers 3x3 512 @25x42", 32, 15, (60, 25, 42, 512), 512, 3, False), ("pers 1x1 1024->256", 32, 15, (60, 50, 84, 1024), 256, 1, False),
         ("stream 256->1024 +res", 30, 20, (60, 50, 84, 256), 1024, 1, True), ("stream 512->256", 30, 20, (60, 100, 168, 512), 256, 1, False),
         ("stream 256->256 +res", 30, 20, (60, 100, 168, 256), 256, 1, True), ("stream 8-wave 512->2048 +res", 30, 20, (60, 25, 42, 512), 2048, 1, True),
         ("halo mf1 3x3 256 b30 P2", 37, 22, (30, 200, 336, 256), 256, 3, False), ("halo mf0 3x3 256 @100x168 +res", 36, 15, (30, 100, 168, 256), 256, 3, True),
         ("halo pers 3x3 256 b30 P2", 38, 22, (30, 200, 336, 256), 256, 3, False), ("halo mf1 3x3 128->512 @41x77", 37, 22, (16, 41, 77, 128), 512, 3, False),
         ("halo64 3x3 64->64 @160x160", 27, 2, (16, 160, 160, 64), 64, 3, False), ("halo64 3x3 128->128 @80x80 +res", 27, 2, (32, 80, 80, 128), 128, 3, True),
         ("halo128 3x3 128->128 @77x45", 11, 2, (8, 77, 45, 128), 128, 3, False)]
for name, v, vref, xs, cout, k, res in CASES:
    cin = xs[3]
    w = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (k * k * cin)) ** 0.5
    pc = nn_ops.pack_conv(w, bias=torch.randn((cout,), generator=g) * 0.1, stride=1, pad=k // 2, relu=True).to(dev)
    x = torch.randn(xs, generator=g).to(torch.bfloat16).to(dev)
    r = torch.randn(xs[:3] + (cout,), generator=g).to(torch.bfloat16).to(dev) if res else None
    if v == 32 and k == 3 and xs[0] * xs[1] * xs[2] >= 400000:
        vref = 22
    ref = nn_ops.conv2d(x, pc, residual=r, variant=vref)
    y = torch.empty_like(ref)
    n_bad = 0
    for i in range(REPS):
        y.fill_(7.0)
        nn_ops.conv2d(x, pc, residual=r, variant=v, out=y)
        if not torch.equal(y, ref):
            n_bad += 1
    bad += n_bad
    print(f"{name}: {REPS} runs, {n_bad} mismatches", flush=True)
# fused RPN head on the HALO form against the linear-tile form
w1 = torch.randn((256, 256, 3, 3), generator=g) * (2.0 / 2304) ** 0.5
pc = nn_ops.pack_conv(w1, bias=torch.randn((256,), generator=g) * 0.1, stride=1, pad=1, relu=True).to(dev)
pc2 = nn_ops.pack_conv(torch.randn((15, 256, 1, 1), generator=g) * 0.05, bias=torch.randn((15,), generator=g) * 0.1).to(dev)
for (n, h, w_) in ((30, 200, 336), (30, 100, 168)):
    x = torch.randn((n, h, w_, 256), generator=g).to(torch.bfloat16).to(dev)
    ref = nn_ops.conv2d_head(x, pc, pc2, variant=35)
    n_bad = sum(0 if torch.equal(nn_ops.conv2d_head(x, pc, pc2, variant=34), ref) else 1 for _ in range(REPS))
    bad += n_bad
    print(f"head halo {n}x{h}x{w_}: {REPS} runs, {n_bad} mismatches", flush=True)
# md_bottleneck (identity / first block) against the three launches
import importlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
for cin, ds in ((256, False), (64, True)):
    w1 = torch.randn((64, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5
    w2 = torch.randn((64, 64, 3, 3), generator=g) * (2.0 / 576) ** 0.5
    w3 = torch.randn((256, 64, 1, 1), generator=g) * (2.0 / 64) ** 0.5
    p1, p2, p3 = (nn_ops.pack_conv(w1, bias=torch.randn((64,), generator=g) * 0.1, relu=True).to(dev),
                  nn_ops.pack_conv(w2, bias=torch.randn((64,), generator=g) * 0.1, stride=1, pad=1, relu=True).to(dev),
                  nn_ops.pack_conv(w3, bias=torch.randn((256,), generator=g) * 0.1, relu=True).to(dev))
    pd = nn_ops.pack_conv(torch.randn((256, cin, 1, 1), generator=g) * (1.0 / cin) ** 0.5, bias=torch.randn((256,), generator=g) * 0.1, relu=False).to(dev) if ds else None
    blk = nn_ops.pack_bottleneck(p1, p2, p3, pd)
    x = torch.randn((16, 200, 336, cin), generator=g).to(torch.bfloat16).to(dev)
    res = nn_ops.conv2d(x, pd) if ds else x
    ref = nn_ops.conv2d(nn_ops.conv2d(nn_ops.conv2d(x, p1), p2), p3, residual=res)
    n_bad = sum(0 if torch.equal(nn_ops.bottleneck(x, blk), ref) else 1 for _ in range(REPS))
    bad += n_bad
    print(f"md_bottleneck Cin {cin} downsample {ds}: {REPS} runs, {n_bad} mismatches", flush=True)
ps = nn_ops.pack_stem_conv(torch.randn((32, 3, 6, 6), generator=g) * 0.1, act="silu").to(dev)
x4 = nn_ops.to_stem_layout(torch.randn((32, 640, 640, 8), generator=g).to(torch.bfloat16).to(dev))
ref = nn_ops.stem_conv(x4, ps)
n_bad = sum(0 if torch.equal(nn_ops.stem_conv(x4, ps), ref) else 1 for _ in range(REPS))
bad += n_bad
print(f"stem_conv 6x6 b32: {REPS} runs, {n_bad} mismatches", flush=True)
# r04: md_c3_pair (three widths) against the two launches, md_sppf_pool against three md_maxpool2d, md_topk_segmented (LDS form) against its first result
for C, H in ((32, 160), (64, 80), (128, 40)):
    bnp = lambda c: (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5, 1e-3)
    q1 = nn_ops.pack_conv(torch.randn((C, C, 1, 1), generator=g) * (2.0 / C) ** 0.5, bn=bnp(C), relu="silu").to(dev)
    q2 = nn_ops.pack_conv(torch.randn((C, C, 3, 3), generator=g) * (2.0 / (9 * C)) ** 0.5, bn=bnp(C), stride=1, pad=1, relu="silu").to(dev)
    pk = nn_ops.pack_c3_pair(q1, q2)
    x = torch.randn((32, H, H, 2 * C), generator=g).to(torch.bfloat16).to(dev)
    ref = nn_ops.conv2d(nn_ops.conv2d(x, q1, x_c_off=0), q2, residual=x, res_c_off=0)
    y = torch.empty_like(x)
    n_bad = 0
    for _ in range(REPS):
        y.fill_(7.0)
        nn_ops.c3_pair(x, pk, y, 0, 0, True, True)
        n_bad += 0 if (torch.equal(y[..., :C], ref) and torch.equal(y[..., C:], x[..., C:])) else 1
    bad += n_bad
    print(f"md_c3_pair C {C} @{H}x{H} b32: {REPS} runs, {n_bad} mismatches", flush=True)
xs_ = torch.randn((32, 20, 20, 256), generator=g).to(torch.bfloat16)
refs = [xs_.to(dev)]
for _ in range(3):
    refs.append(nn_ops.maxpool2d(refs[-1], 5, 1, 2, zero_pad=False))
ref = torch.cat(refs, 3)
n_bad = 0
for _ in range(REPS):
    cat = torch.full((32, 20, 20, 1024), 7.0, dtype=torch.bfloat16, device=dev)
    cat[..., :256] = refs[0]
    nn_ops.sppf_pool(cat, 256, 5)
    n_bad += 0 if torch.equal(cat, ref) else 1
bad += n_bad
print(f"md_sppf_pool 32x20x20x256: {REPS} runs, {n_bad} mismatches", flush=True)
from minddet_amd import det_ops
sc = torch.rand((32 * 25200,), generator=g).to(dev)
seg = torch.arange(0, 33 * 25200, 25200, dtype=torch.int32, device=dev)
v0, i0, c0 = det_ops.topk_segmented(sc, seg, 4096, min_score=0.25, max_segment=25200)
vs, order = torch.sort(sc.view(32, 25200), dim=1, descending=True, stable=True)
assert torch.equal(v0, vs[:, :4096]) and torch.equal(i0.long(), order[:, :4096])
n_bad = 0
for _ in range(REPS):
    v, i, c = det_ops.topk_segmented(sc, seg, 4096, min_score=0.25, max_segment=25200)
    n_bad += 0 if (torch.equal(v, v0) and torch.equal(i, i0) and torch.equal(c, c0)) else 1
bad += n_bad
print(f"md_topk_segmented 32 x 25200, k 4096 (equal to torch.sort(stable) once, then to itself): {REPS} runs, {n_bad} mismatches", flush=True)
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)

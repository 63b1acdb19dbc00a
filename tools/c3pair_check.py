"""md_c3_pair against the two md_conv2d launches it replaces (bit compare) + timing of both, on YOLOv5s' shard shapes.
python tools/c3pair_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops, _lib
if os.environ.get("MD_LIB_OVERRIDE"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MD_LIB_OVERRIDE"])

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
ev = lambda: torch.cuda.Event(enable_timing=True)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[2]


for (N, H, W, C, shortcut, passt) in [(2, 19, 37, 32, True, True), (32, 160, 160, 32, True, True), (2, 19, 37, 64, True, False), (2, 8, 16, 128, True, True), (3, 21, 50, 128, False, True), (1, 5, 3, 64, True, True),
                                     (32, 80, 80, 64, True, False), (32, 40, 40, 128, True, True), (32, 40, 40, 128, False, True), (32, 80, 80, 64, False, True)]:
    bn = lambda c: (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5, 1e-3)
    pc1 = nn_ops.pack_conv(torch.randn((C, C, 1, 1), generator=g) * (2.0 / C) ** 0.5, bn=bn(C), relu="silu").to(dev)
    pc2 = nn_ops.pack_conv(torch.randn((C, C, 3, 3), generator=g) * (2.0 / (9 * C)) ** 0.5, bn=bn(C), stride=1, pad=1, relu="silu").to(dev)
    pk = nn_ops.pack_c3_pair(pc1, pc2)
    assert pk is not None
    cat = torch.randn((N, H, W, 2 * C), generator=g).to(torch.bfloat16).to(dev)
    # unfused, in place on a copy: the graph's own sequence
    ref = cat.clone()
    t = nn_ops.conv2d(ref, pc1, x_c_off=0)
    if shortcut:
        nn_ops.conv2d(t, pc2, residual=ref, res_c_off=0, out=ref, c_off=0)
    else:
        nn_ops.conv2d(t, pc2, out=ref, c_off=0)
    out = torch.full_like(cat, 7.0)
    nn_ops.c3_pair(cat, pk, out, 0, 0, shortcut, passt)
    torch.cuda.synchronize()
    same = torch.equal(out[..., :C], ref[..., :C])
    nd = int((out[..., :C] != ref[..., :C]).sum())
    md = float((out[..., :C].float() - ref[..., :C].float()).abs().max())
    pas = torch.equal(out[..., C:], cat[..., C:]) if passt else bool((out[..., C:] == 7.0).all())
    line = f"N{N} {H}x{W} C{C} shortcut {int(shortcut)} pass {int(passt)}: identical {same} (differ {nd}, max {md:.3g}) second half ok {pas}"
    if N >= 32:
        work = cat.clone()
        def unfused():
            tt = nn_ops.conv2d(work, pc1, x_c_off=0)
            if shortcut:
                nn_ops.conv2d(tt, pc2, residual=work, res_c_off=0, out=work, c_off=0)
            else:
                nn_ops.conv2d(tt, pc2, out=work, c_off=0)
        tu = timed(unfused)
        tf = timed(lambda: nn_ops.c3_pair(cat, pk, out, 0, 0, shortcut, passt))
        line += f" | two launches {tu:6.1f} us, md_c3_pair {tf:6.1f} us"
    print(line, flush=True)

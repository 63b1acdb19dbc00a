"""HALO form of the ping-pong kernel vs the linear-tile form, per layer, interleaved rounds in one process, bit-compare.
python tools/pp_halo_ab.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
LAYERS = [(200, 336, 256, 256), (100, 168, 256, 256), (50, 84, 256, 256), (50, 84, 512, 512), (25, 42, 512, 512)]
for (H, W, Cin, Cout) in LAYERS:
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (9 * Cin)) ** 0.5
    pc = nn_ops.pack_conv(w, bias=torch.randn((Cout,), generator=g) * 0.1, stride=1, pad=1, relu=False).to(dev)
    pc2 = nn_ops.pack_conv(torch.randn((15, 256, 1, 1), generator=g) * 0.05, bias=torch.zeros(15)).to(dev) if Cout == 256 else None
    pcr = nn_ops.pack_conv(w, bias=torch.randn((Cout,), generator=g) * 0.1, stride=1, pad=1, relu=True).to(dev)
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
    y = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=dev)
    fl = 2.0 * B * H * W * Cout * Cin * 9
    arms = [("pp mf0", lambda: nn_ops.conv2d(x, pc, variant=15, out=y)), ("pp mf1", lambda: nn_ops.conv2d(x, pc, variant=22, out=y)),
            ("halo mf0", lambda: nn_ops.conv2d(x, pc, variant=36, out=y)), ("halo mf1", lambda: nn_ops.conv2d(x, pc, variant=37, out=y)),
            ("pers", lambda: nn_ops.conv2d(x, pc, variant=32, out=y)), ("pers halo", lambda: nn_ops.conv2d(x, pc, variant=38, out=y))]
    if pc2 is not None:
        arms += [("head", lambda: nn_ops.conv2d_head(x, pcr, pc2, variant=35)), ("head halo", lambda: nn_ops.conv2d_head(x, pcr, pc2, variant=34))]
    outs = {}
    for nm, fn in arms:
        outs[nm] = fn().clone()
    same = all(torch.equal(outs[n], outs["pp mf0"]) for n in ("pp mf1", "halo mf0", "halo mf1", "pers", "pers halo"))
    if pc2 is not None:
        same = same and torch.equal(outs["head"], outs["head halo"])
    times = {nm: [] for nm, _ in arms}
    for rnd in range(5):
        for nm, fn in arms:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record(); torch.cuda.synchronize()
            times[nm].append(e0.elapsed_time(e1) / 3)
    line = f"{B}x{H}x{W} {Cin}->{Cout}: identical={same}"
    for nm, _ in arms:
        t = sorted(times[nm])[2]
        line += f" | {nm} {t*1e3:7.1f}us {fl/t/1e9:5.0f}TF"
    print(line, flush=True)

"""K = 512 pointwise layers on conv1x1_stream_kernel: 128-cout (4-wave) workgroups (tune stream_cache_bits = 16) against the 256-cout (8-wave) form,
bit-compared, interleaved, a 1.5 GB fill before every timed launch (the layer finds nothing of its operands in the caches, as inside the step).
usage: python tools/stream_nw_ab.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
# H, W, Cin, Cout, residual (0 none, 1 same layout, 2 nearest-2x-upsampled)
LAYERS = [(100, 168, 512, 256, 0), (100, 168, 512, 256, 2), (25, 42, 512, 2048, 1), (50, 84, 512, 1024, 0), (50, 84, 512, 256, 1), (13, 21, 512, 512, 1)]
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
scratch = torch.empty((1536 << 20,), dtype=torch.uint8, device=dev)
narrow, wide = nn_ops.ConvTune(stream_cache_bits=16), nn_ops.ConvTune()
for (H, W, Cin, Cout, res) in LAYERS:
    w = torch.randn((Cout, Cin, 1, 1), generator=g) * (2.0 / Cin) ** 0.5
    pc = nn_ops.pack_conv(w, bias=torch.randn((Cout,), generator=g) * 0.1, relu=res == 1).to(dev)
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
    r = None
    if res == 1:
        r = torch.randn((B, H, W, Cout), generator=g).to(torch.bfloat16).to(dev)
    elif res == 2:
        r = torch.randn((B, (H + 1) // 2, (W + 1) // 2, Cout), generator=g).to(torch.bfloat16).to(dev)
    kw = dict(residual=r, res_upsample=res == 2)
    ref = nn_ops.conv2d(x, pc, variant=20, **kw)
    y = torch.empty_like(ref)
    for tn in (narrow, wide):
        assert torch.equal(nn_ops.conv2d(x, pc, variant=30, tune=tn, **kw), ref) and _lib.lib().md_conv2d_last_kernel() == 8, (H, W, Cin, Cout, res)
    times = {"4 waves": [], "8 waves": []}
    for rnd in range(7):
        for nm, tn in (("4 waves", narrow), ("8 waves", wide)):
            tt = 0.0
            for _ in range(3):
                scratch.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                nn_ops.conv2d(x, pc, variant=30, out=y, tune=tn, **kw)
                e1.record()
                torch.cuda.synchronize()
                tt += e0.elapsed_time(e1)
            times[nm].append(tt / 3)
    by = 2.0 * B * H * W * (Cin + Cout) + (0 if r is None else r.numel() * 2.0)
    line = f"{B}x{H}x{W}x{Cin}->{Cout} k1 res {res}:"
    for nm in times:
        t = sorted(times[nm])[3]
        line += f"  {nm} {t*1e3:7.1f} us {by/t/1e9:5.2f} TB/s"
    print(line, flush=True)

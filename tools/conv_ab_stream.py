"""A/B on the pointwise layers of the ResNet stages: 128x128 single-buffer kernel (variant 20) vs conv1x1_stream_kernel (variant 30),
interleaved rounds in one process.  --flush: a 1.5 GB fill runs before every timed launch, so the layer finds neither its input nor
its residual in the Infinity Cache (as inside the detector's step; without it a replay loop flatters the stream kernel, whose
activation reads are the only cached stream).  python tools/conv_ab_stream.py [batch] [--flush] [--rounds 1,2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops, _lib

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 60
FLUSH = "--flush" in sys.argv
ROUNDS = [int(v) for v in sys.argv[sys.argv.index("--rounds") + 1].split(",")] if "--rounds" in sys.argv else [1]
LAYERS = [(50, 84, 256, 1024, True), (100, 168, 128, 512, True), (25, 42, 512, 2048, True), (100, 168, 512, 128, False),
          (200, 336, 256, 128, False), (100, 168, 512, 256, False), (200, 336, 256, 256, False), (100, 168, 256, 256, True)]
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
lib = _lib.lib()
scratch = torch.empty((1536 << 20,), dtype=torch.uint8, device=dev) if FLUSH else None
for (H, W, Cin, Cout, res) in LAYERS:
    b = B if B * H * W * max(Cin, Cout) * 2 < 0x7fff0000 else B // 2
    w = torch.randn((Cout, Cin, 1, 1), generator=g) * (2.0 / Cin) ** 0.5
    pc = nn_ops.pack_conv(w, relu=True).to(dev)
    x = torch.randn((b, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
    r = torch.randn((b, H, W, Cout), generator=g).to(torch.bfloat16).to(dev) if res else None
    y = torch.empty((b, H, W, Cout), dtype=torch.bfloat16, device=dev)
    by = 2.0 * b * H * W * (Cin + Cout * (2 if res else 1))
    arms = [("v20", 20, 1)] + [(f"s{rd}", 30, rd) for rd in ROUNDS]
    times = {a[0]: [] for a in arms}
    ref = nn_ops.conv2d(x, pc, residual=r, variant=20)
    for (nm, v, rd) in arms:
        assert torch.equal(nn_ops.conv2d(x, pc, residual=r, variant=v, tune=nn_ops.ConvTune(stream_rounds=rd)), ref), (nm, H, W, Cin, Cout)
    for rnd in range(7):
        for (nm, v, rd) in arms:
            tn = nn_ops.ConvTune(stream_rounds=rd)
            tt = 0.0
            for _ in range(3):
                if FLUSH:
                    scratch.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                nn_ops.conv2d(x, pc, residual=r, variant=v, out=y, tune=tn)
                e1.record()
                torch.cuda.synchronize()
                tt += e0.elapsed_time(e1)
            times[nm].append(tt / 3)
    line = f"{b}x{H}x{W}x{Cin}->{Cout} k1{' +res' if res else ''}{' flushed' if FLUSH else ''}:"
    for (nm, v, rd) in arms:
        t = sorted(times[nm])[3]
        line += f"  {nm} {t*1e3:7.1f}us {by/t/1e9:5.2f}TB/s"
    print(line, flush=True)

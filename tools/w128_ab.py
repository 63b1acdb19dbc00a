"""conv_w128_kernel (variant 41: four waves, one per SIMD, a 128 x 128 register tile each) against the ping-pong kernel (15 = 32x32x16, 22 = 16x16x32 MFMA)
on the detector's MFMA-bound layer shapes: bit-compare, then interleaved timing rounds in ONE process on N(0,1) data (guide rule 24 / 25).
The kernel (variant 41) exists in commit "conv_w128_kernel (experiment)" of round 4 only -- measured, not adopted, removed
(profiles/r04_w128_experiment.txt); on later trees variant 41 falls through to the dispatcher's own choice.
python tools/w128_ab.py [rounds] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
VARIANTS = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [15, 22, 41]
ZERO = len(sys.argv) > 4 and sys.argv[4] == "zero"   # all-zero activations and weights: the schedule's rate without the clock the operands cost
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
# name, (N, H, W, Cin), Cout, k, residual
CASES = [("fc 12544->1024 x 60000", (60000, 1, 1, 12544), 1024, 1, False),
         ("3x3 256->256 @200x336 b30", (30, 200, 336, 256), 256, 3, False),
         ("3x3 256->256 @100x168 b60", (60, 100, 168, 256), 256, 3, False),
         ("3x3 256->256 @50x84 b60", (60, 50, 84, 256), 256, 3, False),
         ("3x3 256->256 @40x40 b32 (yolov8l)", (32, 40, 40, 256), 256, 3, False),
         ("1x1 1024->256 @50x84 b60", (60, 50, 84, 1024), 256, 1, False),
         ("1x1 2048->512 @25x42 b60", (60, 25, 42, 2048), 512, 1, False),
         ("3x3 512->512 @25x42 b60", (60, 25, 42, 512), 512, 3, False)]
ev = lambda: torch.cuda.Event(enable_timing=True)
for name, xs, cout, k, res in CASES:
    cin = xs[3]
    w = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (k * k * cin)) ** 0.5
    pc = nn_ops.pack_conv(w, bias=torch.randn((cout,), generator=g) * 0.1, stride=1, pad=k // 2, relu=True).to(dev)
    x = torch.randn(xs, generator=g).to(torch.bfloat16).to(dev)
    if ZERO:
        x.zero_(); pc.w.zero_()
    ref = nn_ops.conv2d(x, pc, variant=15)
    outs = {}
    for v in VARIANTS:
        y = torch.full_like(ref, 7.0)
        nn_ops.conv2d(x, pc, variant=v, out=y)
        torch.cuda.synchronize()
        outs[v] = "bit-identical" if torch.equal(y, ref) else f"DIFFERS in {int((y != ref).sum())} of {ref.numel()} (max {float((y.float() - ref.float()).abs().max()):.3g})"
    flops = 2.0 * ref.numel() * cin * k * k
    times = {v: [] for v in VARIANTS}
    y = torch.empty_like(ref)
    for r in range(ROUNDS):
        for v in VARIANTS:
            nn_ops.conv2d(x, pc, variant=v, out=y)   # warm
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(REPS):
                nn_ops.conv2d(x, pc, variant=v, out=y)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / REPS)
    line = f"{name:36s}"
    for v in VARIANTS:
        t = sorted(times[v])
        med = t[len(t) // 2]
        line += f" | v{v}: {med * 1e3:8.1f} us {flops / med / 1e9:7.1f} TF (min {t[0] * 1e3:.1f}) {outs[v] if v != 15 else ''}"
    print(line, flush=True)

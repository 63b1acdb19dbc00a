"""md_conv1x1_dual (conv3 + strided downsample conv of a stage's first block as one GEMM) on the 128x128 kernel vs the ping-pong kernel
(per-call knob md_conv_tune.dual_pp_min_k), interleaved, bit-compare.  python tools/dual_pp_ab.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 120
lib = _lib.lib()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for (Ho, Wo, Ca, Cb, Cout) in ((100, 168, 128, 256, 512), (50, 84, 256, 512, 1024), (25, 42, 512, 1024, 2048)):
    pc3 = nn_ops.pack_conv(torch.randn((Cout, Ca, 1, 1), generator=g) * (2.0 / Ca) ** 0.5, bias=torch.randn((Cout,), generator=g) * 0.1, relu=True)
    pd = nn_ops.pack_conv(torch.randn((Cout, Cb, 1, 1), generator=g) * (2.0 / Cb) ** 0.5, bias=torch.randn((Cout,), generator=g) * 0.1, stride=2, relu=False)
    pk = nn_ops.pack_dual(pc3, pd)
    pk.w, pk.bias = pk.w.to(dev), pk.bias.to(dev)
    xa = torch.relu(torch.randn((B, Ho, Wo, Ca), generator=g)).to(torch.bfloat16).to(dev)
    xb = torch.relu(torch.randn((B, 2 * Ho, 2 * Wo, Cb), generator=g)).to(torch.bfloat16).to(dev)
    out = torch.empty((B, Ho, Wo, Cout), dtype=torch.bfloat16, device=dev)
    T = {0: nn_ops.ConvTune(dual_pp_min_k=1 << 30), 1: nn_ops.ConvTune(dual_pp_min_k=128)}
    ref = nn_ops.conv1x1_dual(xa, xb, pk, tune=T[0]).clone()
    got = nn_ops.conv1x1_dual(xa, xb, pk, tune=T[1])
    kern = lib.md_conv2d_last_kernel()
    same = torch.equal(ref, got)
    times = {0: [], 1: []}
    for rnd in range(5):
        for arm in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                nn_ops.conv1x1_dual(xa, xb, pk, out=out, tune=T[arm])
            e1.record(); torch.cuda.synchronize()
            times[arm].append(e0.elapsed_time(e1) / 4)
    t0, t1 = sorted(times[0])[2], sorted(times[1])[2]
    fl = 2.0 * B * Ho * Wo * Cout * (Ca + Cb)
    print(f"{B}x{Ho}x{Wo} [{Ca};{Cb}]->{Cout}: 128x128 {t0*1e3:7.1f} us ({fl/t0/1e9:5.0f} TF)  ping-pong {t1*1e3:7.1f} us ({fl/t1/1e9:5.0f} TF) kernel id {kern}  ratio {t0/t1:.3f}  identical={same}", flush=True)

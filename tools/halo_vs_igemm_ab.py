"""3x3 / stride-1 layers with Cout <= 128: the dispatcher's choice (0) against the 128x128 LDS-DMA kernel (2 / 20), the halo-reuse kernel with 64-cout (27)
and 128-cout (11) tiles -- after the r03 lane -> pixel map of the halo kernel.  Interleaved, bit-compared.  usage: python tools/halo_vs_igemm_ab.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
LAYERS = [(100, 168, 128, 128), (50, 84, 128, 128), (200, 336, 64, 64), (100, 168, 64, 128), (25, 42, 128, 128)]
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for (H, W, Cin, Cout) in LAYERS:
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (9 * Cin)) ** 0.5
    pc = nn_ops.pack_conv(w, bias=torch.randn((Cout,), generator=g) * 0.1, stride=1, pad=1, relu=True).to(dev)
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
    ref = nn_ops.conv2d(x, pc, variant=2)
    y = torch.empty_like(ref)
    arms = [("auto", 0), ("igemm 2buf", 2), ("igemm 1buf", 20), ("halo64", 27), ("halo128", 11)]
    kid = {}
    for nm, v in arms:
        assert torch.equal(nn_ops.conv2d(x, pc, variant=v), ref), (nm, H, W)
        kid[nm] = _lib.lib().md_conv2d_last_kernel()
    times = {nm: [] for nm, _ in arms}
    for rnd in range(5):
        for nm, v in arms:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                nn_ops.conv2d(x, pc, variant=v, out=y)
            e1.record(); torch.cuda.synchronize()
            times[nm].append(e0.elapsed_time(e1) / 4)
    fl = 2.0 * B * H * W * Cout * Cin * 9
    line = f"{B}x{H}x{W} {Cin}->{Cout} k3:"
    for nm, _ in arms:
        t = sorted(times[nm])[2]
        line += f"  {nm}[k{kid[nm]}] {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF"
    print(line, flush=True)

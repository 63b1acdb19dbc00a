"""md_c3_pair launch time against the number of tiles (is the second workgroup per CU resident?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
ev = lambda: torch.cuda.Event(enable_timing=True)
for C, H in ((128, 40), (64, 80)):
    pc1 = nn_ops.pack_conv(torch.randn((C, C, 1, 1), generator=g) * (2.0 / C) ** 0.5, bias=torch.zeros(C), relu="silu").to(dev)
    pc2 = nn_ops.pack_conv(torch.randn((C, C, 3, 3), generator=g) * (2.0 / (9 * C)) ** 0.5, bias=torch.zeros(C), stride=1, pad=1, relu="silu").to(dev)
    pk = nn_ops.pack_c3_pair(pc1, pc2)
    for N in (1, 4, 8, 16, 17, 24, 32, 34, 48, 64, 128):
        x = torch.randn((N, H, H, 2 * C), generator=g).to(torch.bfloat16).to(dev)
        y = torch.empty_like(x)
        fn = lambda: nn_ops.c3_pair(x, pk, y, 0, 0, True, True)
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        tiles = N * ((H + 7) // 8) * ((H + 15) // 16)
        print(f"C{C} {H}x{H} N{N:3d}: {tiles:5d} tiles ({tiles / 256:5.2f} per CU): {sorted(ts)[2]:7.1f} us", flush=True)

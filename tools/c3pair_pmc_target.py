"""PMC target: five md_c3_pair launches per width at YOLOv5s' shard shapes (tools/pmc_multi.sh c3pair tools/c3pair_pmc_target.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for C, H in ((32, 160), (64, 80), (128, 40)):
    pc1 = nn_ops.pack_conv(torch.randn((C, C, 1, 1), generator=g) * (2.0 / C) ** 0.5, bias=torch.zeros(C), relu="silu").to(dev)
    pc2 = nn_ops.pack_conv(torch.randn((C, C, 3, 3), generator=g) * (2.0 / (9 * C)) ** 0.5, bias=torch.zeros(C), stride=1, pad=1, relu="silu").to(dev)
    pk = nn_ops.pack_c3_pair(pc1, pc2)
    x = torch.randn((32, H, H, 2 * C), generator=g).to(torch.bfloat16).to(dev)
    y = torch.empty_like(x)
    for _ in range(5):
        nn_ops.c3_pair(x, pk, y, 0, 0, True, True)
torch.cuda.synchronize()

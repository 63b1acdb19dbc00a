"""Run ONE conv layer shape repeatedly (for rocprofv3 --pmc passes). Usage:
python tools/conv_one.py H W Cin Cout k stride batch variant reps [residual 0|1] [head 0|1: md_conv2d_head with a 15-channel 1x1 head]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from minddet_amd import nn_ops

H, W, Cin, Cout, k, s, B, variant, reps = [int(v) for v in sys.argv[1:10]]
use_res = len(sys.argv) > 10 and int(sys.argv[10]) != 0
g = torch.Generator().manual_seed(0)
w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (k * k * Cin)) ** 0.5
pc = nn_ops.pack_conv(w, stride=s, pad=k // 2, relu=True).to("cuda:0")
x = torch.randn((B, H, W, pc.cin), generator=g).to(torch.bfloat16).to("cuda:0")
ho, wo = nn_ops.conv_out_hw(H, W, pc)
r = torch.randn((B, ho, wo, pc.cout), generator=g).to(torch.bfloat16).to("cuda:0") if use_res else None
use_head = len(sys.argv) > 11 and int(sys.argv[11]) != 0
pc2 = nn_ops.pack_conv(torch.randn((15, Cout, 1, 1), generator=g) * 0.05, bias=torch.zeros(15)).to("cuda:0") if use_head else None
for _ in range(reps):
    y = nn_ops.conv2d_head(x, pc, pc2, variant=variant) if use_head else nn_ops.conv2d(x, pc, residual=r, variant=variant)
torch.cuda.synchronize()
print("done", tuple(y.shape))

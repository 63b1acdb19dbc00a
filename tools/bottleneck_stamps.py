"""Diagnostic (-DMD_DIAG build): cycle stamps of one mid-grid workgroup of bottleneck64_kernel -- where a tile's lifetime goes.
Usage: python tools/bottleneck_stamps.py [batch H W Cin]"""
import ctypes
import os
import subprocess
import sys

os.environ["MD_DIAG_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from minddet_amd import _lib, nn_ops

subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc"), "diag", "-j8"])
B, H, W, Cin = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (60, 200, 336, 256)
dev = "cuda:0"
stamps = torch.zeros(16, dtype=torch.int64, device=dev)
_lib.lib().md_diag_set_bn_stamp_buffer(ctypes.c_void_p(stamps.data_ptr()))
g = torch.Generator().manual_seed(0)
pc1 = nn_ops.pack_conv(torch.randn((64, Cin, 1, 1), generator=g) * (2.0 / Cin) ** 0.5, bias=torch.zeros(64), relu=True).to(dev)
pc2 = nn_ops.pack_conv(torch.randn((64, 64, 3, 3), generator=g) * (2.0 / 576) ** 0.5, bias=torch.zeros(64), stride=1, pad=1, relu=True).to(dev)
pc3 = nn_ops.pack_conv(torch.randn((256, 64, 1, 1), generator=g) * (2.0 / 64) ** 0.5, bias=torch.zeros(256), relu=True).to(dev)
pd = nn_ops.pack_conv(torch.randn((256, Cin, 1, 1), generator=g) * 0.1, bias=torch.zeros(256), relu=False).to(dev) if Cin == 64 else None
blk = nn_ops.pack_bottleneck(pc1, pc2, pc3, pd)
x = torch.relu(torch.randn((B, H, W, Cin), generator=torch.Generator(device=dev).manual_seed(1), device=dev)).to(torch.bfloat16)
res = None
for _ in range(5):
    nn_ops.bottleneck(x, blk, residual=res)
torch.cuda.synchronize()
st = stamps.cpu().tolist()
names = ["first chunk landed", "phase A loop done", "T1 written + taps 0-4 landed", "taps 0-4 multiplied", "taps 5-8 landed", "taps 5-8 multiplied",
         "T2 written + W3 landed", "quarter 0", "quarter 1", "quarter 2", "quarter 3", "stores drained"]
print(f"{B}x{H}x{W}x{Cin}: workgroup lifetime {st[12] - st[0]} cycles")
if st[10] == 0:   # the barrier-free phase C of the identity / residual-tensor blocks (r03): two quarter PAIRS per wave, stamps 8 and 9 only
    names = names[:7] + ["quarters 2wc (wave-private slab)", "quarters 2wc + 1"]
    for i, n_ in enumerate(names):
        print(f"  {n_:34s} {st[i + 1] - st[i]:7d}")
    print(f"  {'stores drained':34s} {st[12] - st[9]:7d}")
    print(f"  first quarter in detail: MFMAs + bias/pack {st[13] - st[7]}, slab write / read-out / residual add / stores {st[8] - st[13]}")
else:
    for i, n_ in enumerate(names):
        print(f"  {n_:34s} {st[i + 1] - st[i]:7d}")
    print(f"  quarter 1 in detail: MFMAs {st[13] - st[8]}, image write + barrier {st[14] - st[13]}, residual add + store issue {st[15] - st[14]}, barrier {st[9] - st[15]}")

"""Diagnostic: s_memtime stamps of one mid-grid workgroup of the 128x128 conv kernel (variant 25 of the -DMD_DIAG build; the stamps go to a buffer of
their own).  Sections: setup | first DMA wait | K loop | residual issue + LDS transpose | barrier | store issue |
store drain.  Usage: python tools/igemm_stamps.py H W Cin Cout k res(0/1)"""
import ctypes
import os
import subprocess
import sys

os.environ["MD_DIAG_LIB"] = "1"   # the -DMD_DIAG build (make -C minddet_amd/csrc diag); the product library rejects variant 25
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from minddet_amd import _lib, nn_ops

subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc"), "diag", "-j8"])
stamps = torch.zeros(8 * 16, dtype=torch.int64, device="cuda:0")
_lib.lib().md_diag_set_stamp_buffer(ctypes.c_void_p(stamps.data_ptr()))

H, W, Cin, Cout, k, res = [int(v) for v in sys.argv[1:7]] if len(sys.argv) > 6 else (200, 336, 64, 256, 1, 1)
g = torch.Generator().manual_seed(0)
w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (k * k * Cin)) ** 0.5
pc = nn_ops.pack_conv(w, stride=1, pad=k // 2, relu=True, korder=0).to("cuda:0")
x = torch.randn((32, H, W, Cin), generator=g).to(torch.bfloat16).to("cuda:0")
r = torch.randn((32, H, W, Cout), generator=g).to(torch.bfloat16).to("cuda:0") if res else None
for _ in range(5):
    y = nn_ops.conv2d(x, pc, residual=r, variant=25)
torch.cuda.synchronize()
st = stamps.cpu()[:7].tolist()
names = ["setup", "dma wait", "k loop", "res+transpose", "barrier", "store issue", "store drain"]
d = [st[i + 1] - st[i] for i in range(6)]
print(f"{H}x{W}x{Cin}->{Cout} k{k} res={res}: lifetime {st[6] - st[0]} cycles")
for n_, v in zip(names[1:], d):
    print(f"  {n_:16s} {v:7d}")

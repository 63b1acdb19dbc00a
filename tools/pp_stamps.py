"""Diagnostic: per-section s_memtime stamps of the ping-pong conv kernel (variant 19 = stamp build, wrong output).
Prints, for the last K tile of workgroup 0, the cycles each wave spent in: reads+DMA issue | waits | barrier A |
MFMA cluster | barrier B, for both phases.  Usage: python tools/pp_stamps.py [H W Cin Cout k]"""
import os
import subprocess
import sys

os.environ["MD_DIAG_LIB"] = "1"   # the -DMD_DIAG build (make -C minddet_amd/csrc diag); the product library rejects variant 19
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from minddet_amd import _lib, nn_ops

subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc"), "diag", "-j8"])
stamps = torch.zeros(8 * 16, dtype=torch.int64, device="cuda:0")
_lib.lib().md_diag_set_stamp_buffer(__import__("ctypes").c_void_p(stamps.data_ptr()))

H, W, Cin, Cout, k = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (200, 336, 256, 256, 3)
VAR = int(sys.argv[6]) if len(sys.argv) > 6 else 19   # 19: 32x32x16 MFMA, 26: 16x16x32 MFMA
g = torch.Generator().manual_seed(0)
w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (k * k * Cin)) ** 0.5
pc = nn_ops.pack_conv(w, stride=1, pad=k // 2, relu=True, korder=1 if k > 1 else 0).to("cuda:0")
x = torch.randn((32 if H > 1 else 8, H, W, Cin), generator=g).to(torch.bfloat16).to("cuda:0")
import time
t_end = time.time() + 2.5   # >= 2 s of back-to-back launches on random data before the stamps are read (MICROARCH DVFS item 6)
while time.time() < t_end:
    for _ in range(20):
        y = nn_ops.conv2d(x, pc, variant=VAR)
    torch.cuda.synchronize()
torch.cuda.synchronize()
raw = stamps.cpu().reshape(8, 16)
st = raw[:, :11]
nk = k * k * Cin // 64
print(f"variant {VAR} | main loop: {int(raw[0, 11])} cycles = {int(raw[0, 11]) / nk:.0f} per K tile, in-kernel clock "
      f"{float(raw[0, 11]) / float(raw[0, 12]) * 100:.0f} MHz; prologue {int(raw[0, 13])} cycles, epilogue {int(raw[0, 14])} cycles")
names = ["issue0", "wait0", "barA0", "mfma0", "barB0", "issue1", "wait1", "barA1", "mfma1", "barB1"]
print("wave " + " ".join(f"{n:>7s}" for n in names) + "   total   start-offset")
t0 = int(st[:, 0].min())
for wv in range(8):
    d = [int(st[wv, i + 1] - st[wv, i]) for i in range(10)]
    print(f"{wv:4d} " + " ".join(f"{v:7d}" for v in d) + f" {sum(d):7d}   {int(st[wv, 0]) - t0:7d}")

"""Diagnostic (-DMD_DIAG build): cycle stamps of one mid-grid workgroup of c3pair64_kernel / c3pair128_kernel -- where a tile's lifetime goes.
Usage: python tools/c3pair_stamps.py [batch H W C]"""
import ctypes, os, subprocess, sys
os.environ["MD_DIAG_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import _lib, nn_ops

subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc"), "diag", "-j8"])
B, H, W, C = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (32, 40, 40, 128)
dev = "cuda:0"
stamps = torch.zeros(16, dtype=torch.int64, device=dev)
_lib.lib().md_diag_set_c3_stamp_buffer(ctypes.c_void_p(stamps.data_ptr()))
g = torch.Generator().manual_seed(0)
pc1 = nn_ops.pack_conv(torch.randn((C, C, 1, 1), generator=g) * (2.0 / C) ** 0.5, bias=torch.zeros(C), relu="silu").to(dev)
pc2 = nn_ops.pack_conv(torch.randn((C, C, 3, 3), generator=g) * (2.0 / (9 * C)) ** 0.5, bias=torch.zeros(C), stride=1, pad=1, relu="silu").to(dev)
pk = nn_ops.pack_c3_pair(pc1, pc2)
x = torch.randn((B, H, W, 2 * C), generator=g).to(torch.bfloat16).to(dev)
y = torch.empty_like(x)
for _ in range(5):
    nn_ops.c3_pair(x, pk, y, 0, 0, True, True)
torch.cuda.synchronize()
st = stamps.cpu().tolist()
names = ["x tile + first weights landed (first barrier)", "phase A MFMAs", "T1 barrier + bias, SiLU, T1 written", "to the first phase-B barrier", "phase B (all sub-units)",
         "b2 read + barrier", "SiLU + image written + barrier", "read-out, shortcut add, stores issued", "stores drained"]
print(f"{B}x{H}x{W} C{C}: workgroup lifetime {st[9] - st[0]} cycles (100 MHz counter x clock ratio: see bottleneck_stamps.py)")
for i, n_ in enumerate(names):
    print(f"  {n_:48s} {st[i + 1] - st[i]:7d}")

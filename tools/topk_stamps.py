"""Diagnostic (-DMD_DIAG build): cycle stamps of segment 0's workgroup of md_topk_segmented."""
import ctypes, os, subprocess, sys
os.environ["MD_DIAG_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import _lib, det_ops
subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc"), "diag", "-j8"])
dev = "cuda:0"
B, n, k, thr = [int(v) for v in sys.argv[1:4]] + [float(sys.argv[4])] if len(sys.argv) > 4 else (32, 25200, 4096, 0.25)
stamps = torch.zeros(16, dtype=torch.int64, device=dev)
assert _lib.lib().md_diag_set_topk_stamp_buffer(ctypes.c_void_p(stamps.data_ptr())) == 0
sc = torch.rand((B * n,), generator=torch.Generator().manual_seed(0)).to(dev)
seg = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev)
for _ in range(3):
    v, i, c = det_ops.topk_segmented(sc, seg, k, min_score=thr, max_segment=n)
torch.cuda.synchronize()
st = stamps.cpu().tolist()
names = ["keys staged (registers + LDS)", "selectable counted", "radix select (4 passes)", "compaction (+ ties)", "sort", "outputs written"]
print(f"{B} x {n}, k {k}: selected {int(c[0])}; lifetime {st[6] - st[0]} cycles")
for j, nm in enumerate(names):
    print(f"  {nm:32s} {st[j + 1] - st[j]:8d}")

"""Evidence that the MFMA-bound conv layers are power-bound: the SAME launch (3x3 256->256 on 200x336, ping-pong kernel) timed with
operands that toggle the matrix pipe differently -- random activations and weights, post-ReLU-like activations (half zeros), all-zero
activations, all-zero activations AND weights.  Identical instruction stream and memory traffic; only the data differs.
python tools/mfma_data_dependence.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
H, W, C = 200, 336, 256
w = torch.randn((C, C, 3, 3), generator=g) * (2.0 / (9 * C)) ** 0.5
pc = nn_ops.pack_conv(w, stride=1, pad=1, relu=True).to(dev)
pc0 = nn_ops.pack_conv(torch.zeros_like(w), stride=1, pad=1, relu=True).to(dev)
x = torch.randn((B, H, W, C), generator=g).to(torch.bfloat16).to(dev)
fl = 2.0 * B * H * W * C * C * 9


def t(xx, p):
    for _ in range(3):
        nn_ops.conv2d(xx, p)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            nn_ops.conv2d(xx, p)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    return sorted(ts)[2]


for name, xx, p in (("random activations, random weights", x, pc), ("post-ReLU activations (half zeros), random weights", torch.relu(x), pc),
                    ("zero activations, random weights", torch.zeros_like(x), pc), ("zero activations, zero weights", torch.zeros_like(x), pc0),
                    ("random activations, random weights (again)", x, pc)):
    ms = t(xx, p)
    print(f"{name:52s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.0f} TFLOP/s", flush=True)

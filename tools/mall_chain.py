"""Does a producer -> consumer pair of HBM-bound layers run faster per image when the tensor between them fits the 256 MB Infinity Cache?
A ResNet stage-3 identity block chain (1x1 1024->256, 3x3 256->256, 1x1 256->1024 + residual) x BLOCKS at 50 x 84, run as B / S sub-batches of S
images: each sub-batch walks ALL blocks before the next starts (depth-first), so that a block's 1024-channel output (S x 8.6 MB) is re-read
while it may still be cache-resident.  Reported: ms per image of the whole chain, per sub-batch size.
python tools/mall_chain.py [B] [blocks] [S,S,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
BLOCKS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
SS = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [60, 30, 20, 15, 10, 6]
H, W, C, M = (int(v) for v in os.environ.get("SHAPE", "50,84,1024,256").split(","))
dev = "cuda:0"
g = torch.Generator().manual_seed(0)


def pack(cout, cin, k, relu=True):
    w = torch.randn((cout, cin, k, k), generator=g) * (1.0 / (k * k * cin)) ** 0.5
    return nn_ops.pack_conv(w, bias=torch.randn((cout,), generator=g) * 0.05, stride=1, pad=k // 2, relu=relu).to(dev)


blocks = [(pack(M, C, 1), pack(M, M, 3), pack(C, M, 1)) for _ in range(BLOCKS)]
x = torch.randn((B, H, W, C), generator=g).to(torch.bfloat16).to(dev)
flush = torch.empty((1536 << 20,), dtype=torch.uint8, device=dev)


def chain(xs):
    for c1, c2, c3 in blocks:
        t = nn_ops.conv2d(xs, c1)
        t = nn_ops.conv2d(t, c2)
        xs = nn_ops.conv2d(t, c3, residual=xs)
    return xs


def run(S):
    outs = [chain(x[i:i + S]) for i in range(0, B, S)]
    return outs


ref = torch.cat(run(B), 0)
for S in SS:
    if B % S:
        continue
    same = torch.equal(torch.cat(run(S), 0), ref)
    ts = []
    for rd in range(5):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(S); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{BLOCKS} blocks {C}/{M} @{H}x{W}, batch {B} as sub-batches of {S:3d} (block output {S * H * W * C * 2 / 2**20:6.1f} MB): median {ts[2]:7.3f} ms = "
          f"{ts[2] / B * 1e3:7.1f} us per image (min {ts[0]:7.3f}) identical: {same}", flush=True)

"""Two questions about the HBM-bound front of the ResNet (stem + stage 1 + stage 2) at the 60-image shard:
 (A) launch size: the same tensors, each launch cut into image chunks by md_conv_tune.chunk_limit (layer by layer: no cache residency)
 (B) depth first: stem + stages run chunk by chunk on separate tensors, so a block's output may still sit in the 256 MiB Infinity Cache when the next reads it
Per-stage times from events inside one pass.  usage: python tools/chunk_residency.py [images]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import graphs, nn_ops
from minddet_amd.data import synthetic_images

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda:0"
net = graphs.ResNet(50).to(dev)
x = nn_ops.to_stem_layout(synthetic_images(N, 800, 1344, device=dev))
net(x[:2])   # packs the fused blocks
torch.cuda.synchronize()


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(fn, reps=4):
    best = None
    for _ in range(reps):
        marks = fn()
        torch.cuda.synchronize()
        t = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)]
        best = t if best is None else [min(a, b) for a, b in zip(best, t)]
    return best


# (A) one identity block of stage 1 / the stage-2 layers on the full tensors, launch cut by chunk_limit
y0 = nn_ops.stem_pool(x, net.stem)
y1 = net.stages[0][0](y0)
blk = net.stages[0][1]
out = torch.empty_like(y1)
per_img = y1[0].numel() * 2
for imgs in (N, N // 2, N // 3, N // 4, N // 6):
    tune = nn_ops.ConvTune(chunk_limit=per_img * imgs + 4096)
    def f():
        m = [ev(), ev()]
        m[0].record(); nn_ops.bottleneck(y1, blk._fused, out=out, tune=tune); m[1].record()
        return m
    print("(A) identity block of stage 1, %2d images per launch: %.3f ms" % (imgs, timed(f)[0]), flush=True)
del out

# (B) depth first
def depth_first(chunk):
    m = [ev()]
    m[0].record()
    outs = []
    for i in range(0, N, chunk):
        y = nn_ops.stem_pool(x[i:i + chunk], net.stem)
        for st in net.stages[:2]:
            for b in st:
                y = b(y)
        outs.append(y)
    m.append(ev()); m[1].record()
    return m

def layer_first():
    m = [ev()]
    m[0].record()
    y = nn_ops.stem_pool(x, net.stem)
    m.append(ev()); m[-1].record()
    for st in net.stages[:2]:
        for b in st:
            y = b(y)
        m.append(ev()); m[-1].record()
    return m

t = timed(layer_first)
print("(B) layer first, %d images: stem %.3f  stage 1 %.3f  stage 2 %.3f  total %.3f ms" % (N, t[0], t[1], t[2], sum(t)), flush=True)
for chunk in (N // 2, N // 3, N // 5, N // 10, N // 15, N // 20):
    if chunk and N % chunk == 0:
        print("(B) depth first, chunk %2d images: total %.3f ms" % (chunk, timed(depth_first.__get__(chunk) if False else (lambda c=chunk: depth_first(c)))[0]), flush=True)
t = timed(layer_first)
print("(B) layer first again: stem %.3f  stage 1 %.3f  stage 2 %.3f  total %.3f ms" % (t[0], t[1], t[2], sum(t)), flush=True)

"""Does running the two halves of a batch on two HIP streams (same kernels, same results per image) hide the launch tails (last partial round of
workgroups, completion skew) of the layer-by-layer path?  Whole detector forward: one stream x N images  vs  2 streams x N/2 images.
usage: python tools/two_stream_halves.py <config> <batch> [streams]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet.models import Config, build_detector
from minddet_amd import nn_ops
from minddet_amd.data import synthetic_images

cfgp, N = sys.argv[1], int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = "cuda:0"
cfg = Config.fromfile(cfgp)
m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev)
hw = (800, 1344) if "rcnn" in cfgp else (640, 640)
x = synthetic_images(N, hw[0], hw[1], device=dev)
x = nn_ops.to_stem_layout(x) if hasattr(nn_ops, "to_stem_layout") and nn_ops.stem_layout_ok(*hw) else x
streams = [torch.cuda.Stream() for _ in range(S)]
parts = [x[i * (N // S):(i + 1) * (N // S)] for i in range(S)]


def one():
    return m.forward(x)


def multi():
    cur = torch.cuda.current_stream()
    outs = []
    for s, p in zip(streams, parts):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(m.forward(p))
    for s in streams:
        cur.wait_stream(s)
    return outs


def timed(fn, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


o1 = one(); o2 = multi(); torch.cuda.synchronize()
same = all(torch.equal(torch.cat([o[k] for o in o2]), o1[k]) for k in range(2))
print("outputs identical:", same)
for r in range(2):
    print("%s b%d: one stream %.3f ms/step | %d streams x %d images %.3f ms/step" % (os.path.basename(cfgp), N, timed(one), S, N // S, timed(multi)), flush=True)

"""YOLO steps as a replayed HIP graph, with one stream and with graphs.SplitForward's two streams captured inside the graph (fork / join in the capture).
usage: python tools/graph_streams_try.py <config> <batch>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet.models import Config, build_detector
from minddet_amd import nn_ops
from minddet_amd.data import synthetic_images
from minddet_amd.graphs import SplitForward
from minddet_amd.replay import CapturedStep

cfgp, N = sys.argv[1], int(sys.argv[2])
dev = "cuda:0"
cfg = Config.fromfile(cfgp)
m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev)
hw = (800, 1344) if "rcnn" in cfgp else (640, 640)
x = synthetic_images(N, hw[0], hw[1], device=dev)
if nn_ops.stem_layout_ok(*hw):
    x = nn_ops.to_stem_layout(x)


def timed(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ref = m.forward(x)
torch.cuda.synchronize()
res = {"eager, 1 stream": timed(lambda: m.forward(x))}
sp = SplitForward(m, 2)
res["eager, 2 streams"] = timed(lambda: sp(x))
g1 = CapturedStep(lambda xx: tuple(m.forward(xx))[:2], x)
res["graph, 1 stream"] = timed(lambda: g1(x))
o = g1(x); torch.cuda.synchronize()
print("graph 1 stream identical:", all(torch.equal(a, b) for a, b in zip(o, ref)))
for parts in (2, 4):
    try:
        spn = SplitForward(m, parts)
        spn(x); torch.cuda.synchronize()
        g2 = CapturedStep(lambda xx: tuple(spn(xx))[:2], x)
        res["graph, %d streams" % parts] = timed(lambda: g2(x))
        o = g2(x); torch.cuda.synchronize()
        print("graph %d streams identical:" % parts, all(torch.equal(a, b) for a, b in zip(o, ref)))
    except Exception as e:   # noqa: BLE001
        print("graph with %d streams failed:" % parts, repr(e)[:300])
for k, v in res.items():
    print("%s b%d %-18s %.3f ms/step" % (os.path.basename(cfgp), N, k, v), flush=True)

"""Try capturing one detector step in a HIP graph (torch.cuda.CUDAGraph) and replaying it.  python tools/graph_try.py [batch] [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet.models import Config, build_detector
from minddet_amd import nn_ops
from minddet_amd.data import synthetic_images

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
cfg = Config.fromfile(sys.argv[2] if len(sys.argv) > 2 else "configs/faster_rcnn/faster_rcnn_r50_fpn.py")
model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev)
H, W = cfg.data.input_hw
images = synthetic_images(B, H, W, device=dev)
if type(model).__name__ in ("FasterRCNN", "MaskRCNN"):
    images = nn_ops.to_stem_layout(images)
for _ in range(3):
    dets, count = model.forward(images)[:2]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    dets, count = model.forward(images)[:2]
torch.cuda.synchronize()
t_eager = (time.perf_counter() - t0) / 10
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        model.forward(images)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gd, gc = model.forward(images)[:2]
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
print("equal:", torch.equal(gd, dets), torch.equal(gc, count))
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / 10
print(f"B={B} eager {t_eager*1e3:.3f} ms  graph {t_graph*1e3:.3f} ms")

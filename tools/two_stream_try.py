"""Experiment: the batch split over two HIP streams (each half runs the whole path) vs one stream.
Idea: tails / HBM-bound launches of one half fill behind the MFMA-bound launches of the other.  Result recorded in DESIGN.md section 6."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minddet.models import Config, build_detector
from minddet_amd import nn_ops
from minddet_amd.data import synthetic_images

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
steps = 8
dev = torch.device("cuda:0")
cfg = Config.fromfile("configs/faster_rcnn/faster_rcnn_r50_fpn.py")
model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev)
H, W = cfg.data.input_hw
x = nn_ops.to_stem_layout(synthetic_images(B, H, W, seed=1, device=dev))


def run(parts, offset_ms=0.0):
    chunks = list(torch.chunk(x, parts, 0))
    streams = [torch.cuda.Stream() for _ in range(parts)]
    outs = [None] * parts
    def one_step():
        for i, (c, s) in enumerate(zip(chunks, streams)):
            with torch.cuda.stream(s):
                outs[i] = model.forward(c)
    for _ in range(3):
        one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dt, outs


ref_dt, ref = run(1)
print(f"1 stream : {ref_dt*1e3:.2f} ms/step  {B/ref_dt:.1f} images/s", flush=True)
for parts in (2, 3):
    if B % parts:
        continue
    dt, outs = run(parts)
    d = torch.cat([o[0] for o in outs], 0)
    same = torch.equal(d, ref[0][0])
    print(f"{parts} streams: {dt*1e3:.2f} ms/step  {B/dt:.1f} images/s   detections identical: {same}", flush=True)

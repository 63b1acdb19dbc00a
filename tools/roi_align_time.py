"""md_roi_align on the detector's own proposals (Faster R-CNN R50-FPN, synthetic batch): time per launch, interleaved with nothing -- run it on
two trees (tools/prof_two_trees.sh style) or before / after a change.  usage: python tools/roi_align_time.py [batch]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from minddet.models import Config, build_detector
from minddet_amd import det_ops, nn_ops
from minddet_amd.data import synthetic_images

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda:0"
cfg = Config.fromfile("configs/faster_rcnn/faster_rcnn_r50_fpn.py")
m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev)
x = nn_ops.to_stem_layout(synthetic_images(B, 800, 1344, seed=20240317, device=dev))
dets, count, aux = m.forward(x, return_aux=True)
feats, rois = aux["feats"], aux["rois"]
h = m.roi_head
r5 = rois if rois.dim() == 2 else rois.reshape(-1, rois.shape[-1])
print("rois", tuple(rois.shape), "levels", [tuple(f.shape) for f in feats[:len(h.strides)]])
scratch = torch.empty((1024 << 20,), dtype=torch.uint8, device=dev)
ts = []
for i in range(8):
    scratch.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = det_ops.roi_align(list(feats[:len(h.strides)]), r5, h.P, [1.0 / s for s in h.strides], h.sampling, True)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("md_roi_align b%d: %s  median %.3f ms" % (B, " ".join("%.3f" % t for t in ts), sorted(ts)[len(ts) // 2]))

"""In-situ tuning of conv1x1_stream_kernel: whole Faster R-CNN steps, one knob setting applied to ONE layer family at a time
(the others keep the defaults).  python tools/stream_insitu_tune.py [batch] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet.models import Config, build_detector
from minddet_amd import nn_ops, _lib
from minddet_amd.data import synthetic_images

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
cfg = Config.fromfile("configs/faster_rcnn/faster_rcnn_r50_fpn.py")
model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev)
H, W = cfg.data.input_hw
x = nn_ops.to_stem_layout(synthetic_images(B, H, W, seed=1, device=dev))
orig = nn_ops.conv2d
lib = _lib.lib()
cur = {"fam": None, "knobs": (1, 2, 6), "variant": None}

def patched(xx, pc, residual=None, relu=None, out=None, variant=None, **kw):
    hit = pc.kh == 1 and (pc.cin, pc.cout) == cur["fam"]
    if hit:
        r, w, t = cur["knobs"]
        kw["tune"] = nn_ops.ConvTune(stream_rounds=r, stream_wgs_per_cu=w, stream_cache_bits=8 | t)   # per-call knobs (md_conv_tune)
        variant = cur["variant"]
    return orig(xx, pc, residual=residual, relu=relu, out=out, variant=variant, **kw)

nn_ops.conv2d = patched
for _ in range(2):
    model.forward(x)
torch.cuda.synchronize()

def run():
    model.forward(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(STEPS):
        model.forward(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / STEPS

SETTINGS = [("default r1 w2 t6", (1, 2, 6), None), ("off (v31)", (1, 2, 6), 31), ("r2", (2, 2, 6), None), ("r4", (4, 2, 6), None),
            ("w3", (1, 3, 6), None), ("w3 r2", (2, 3, 6), None), ("t7 x nt", (1, 2, 7), None), ("t2 st cached", (1, 2, 2), None),
            ("t0 all cached", (1, 2, 0), None), ("t4 res cached", (1, 2, 4), None)]
for fam in [(128, 512), (256, 1024), (512, 256), (512, 2048)]:
    res = {s[0]: [] for s in SETTINGS}
    for rnd in range(3):
        for name, knobs, var in SETTINGS:
            cur.update(fam=fam, knobs=knobs, variant=var)
            res[name].append(run())
    base = sorted(res[SETTINGS[0][0]])[1]
    print(f"== {fam[0]}->{fam[1]}")
    for name, _, _ in SETTINGS:
        t = sorted(res[name])[1]
        print(f"   {name:18s} {t:7.3f} ms/step ({t - base:+.3f})   " + " ".join(f"{v:.3f}" for v in res[name]), flush=True)

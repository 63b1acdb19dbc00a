"""In-situ A/B of conv1x1_stream_kernel: whole Faster R-CNN steps with the stream kernel on (auto), off everywhere (variant 31), and off
for one (Cin, Cout) layer family at a time -- the layer's neighbours, cache state and clocks are the benchmark's, unlike a replay loop.
python tools/stream_insitu_ab.py [batch] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet.models import Config, build_detector
from minddet_amd import nn_ops
from minddet_amd.data import synthetic_images

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
cfg = Config.fromfile("configs/faster_rcnn/faster_rcnn_r50_fpn.py")
model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev)
H, W = cfg.data.input_hw
x = nn_ops.to_stem_layout(synthetic_images(B, H, W, seed=1, device=dev))
orig = nn_ops.conv2d
off = set()
seen = set()

def patched(xx, pc, residual=None, relu=None, out=None, variant=None, **kw):
    if pc.kh == 1 and pc.cin in (256, 512) and pc.cout % 256 == 0:
        seen.add((pc.cin, pc.cout, residual is not None, bool(kw.get("res_upsample"))))
        if "all" in off or (pc.cin, pc.cout) in off:
            variant = 31
    return orig(xx, pc, residual=residual, relu=relu, out=out, variant=variant, **kw)

nn_ops.conv2d = patched
for _ in range(2):
    model.forward(x)
torch.cuda.synchronize()
fams = sorted({(c, o) for (c, o, _, _) in seen})
arms = [("stream on", set())] + [("stream off", {"all"})] + [(f"off only {c}->{o}", {(c, o)}) for (c, o) in fams]
res = {a[0]: [] for a in arms}
for rnd in range(3):
    for name, s in arms:
        off.clear(); off.update(s)
        model.forward(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(STEPS):
            model.forward(x)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / STEPS)
print("layers seen (cin, cout, residual, upsampled):", sorted(seen))
base = sorted(res["stream on"])[1]
for name, _ in arms:
    t = sorted(res[name])[1]
    print(f"{name:24s} {t:7.3f} ms/step  ({t - base:+.3f} ms vs stream on)  rounds: " + " ".join(f"{v:.3f}" for v in res[name]), flush=True)

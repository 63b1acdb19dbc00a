"""Audit of the conv dispatcher on a model's real launches: one forward pass records every md_conv2d call (tensors, packed layer,
residual kind), then each distinct call is replayed with the auto choice (variant 0) and with pinned kernels; prints the calls where
a pinned kernel beats the auto choice by more than 3 %.  python tools/dispatch_audit.py [config] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet.models import Config, build_detector
from minddet_amd import _lib, nn_ops
from minddet_amd.data import synthetic_images

cfg_path = sys.argv[1] if len(sys.argv) > 1 else "configs/faster_rcnn/faster_rcnn_r50_fpn.py"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
cfg = Config.fromfile(cfg_path)
model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev)
H, W = cfg.data.input_hw
x = synthetic_images(B, H, W, seed=1, device=dev)
if type(model).__name__ in ("FasterRCNN", "MaskRCNN") and nn_ops.stem_layout_ok(H, W):
    x = nn_ops.to_stem_layout(x)

calls = {}
orig = nn_ops.conv2d
def rec(xx, pc, residual=None, relu=None, out=None, variant=None, c_off=0, res_upsample=False, x_c_off=None, res_c_off=None):
    y = orig(xx, pc, residual=residual, relu=relu, out=out, variant=variant, c_off=c_off, res_upsample=res_upsample, x_c_off=x_c_off, res_c_off=res_c_off)
    key = (tuple(xx.shape), pc.cin, pc.cout, pc.kh, pc.stride, pc.pad, int(pc.relu if relu is None else relu), residual is not None, bool(res_upsample),
           out is not None, c_off, x_c_off, res_c_off)
    if key not in calls:
        calls[key] = dict(n=0, args=(xx, pc, residual, relu, None if out is None else out, c_off, res_upsample, x_c_off, res_c_off))
    calls[key]["n"] += 1
    return y
nn_ops.conv2d = rec
model.forward(x)
nn_ops.conv2d = orig
torch.cuda.synchronize()
last = _lib.lib().md_conv2d_last_kernel
KN = {1: "pingpong", 2: "igemm128", 3: "small-cout", 4: "generic-K", 5: "halo", 6: "other"}

def timeit(f):
    for _ in range(2):
        f()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    return sorted(ts)[2] * 1e3

tot_auto = tot_best = 0.0
for key, c in sorted(calls.items(), key=lambda kv: -kv[1]["n"]):
    xx, pc, residual, relu, out, c_off, res_up, xo, ro = c["args"]
    korder = getattr(pc, "korder", 0)
    variants = [0, 20, 2]
    if pc.cout % 256 == 0 and pc.cin % 64 == 0:
        variants.append(15)
    if pc.kh == 3 and pc.stride == 1 and korder == 1 and pc.cout % 64 == 0 and pc.cin % 64 == 0:
        variants.append(27)
    res = {}
    for v in variants:
        try:
            f = lambda v=v: orig(xx, pc, residual=residual, relu=relu, out=out, variant=v, c_off=c_off, res_upsample=res_up, x_c_off=xo, res_c_off=ro)
            f()
            k = last()
            res[v] = (timeit(f), KN.get(k, str(k)))
        except _lib.MindDetHipError:
            pass
    t0 = res[0][0]
    best_v = min(res, key=lambda v: res[v][0])
    tot_auto += t0 * c["n"]
    tot_best += res[best_v][0] * c["n"]
    flag = "  <-- auto loses %.0f %%" % (100 * (t0 / res[best_v][0] - 1)) if t0 > 1.03 * res[best_v][0] else ""
    print(f"x{c['n']:2d} {key[0]} cin{key[1]}->{key[2]} k{key[3]}s{key[4]} res={int(key[7])}{'up' if key[8] else ''}: " +
          "  ".join(f"v{v}:{t:.0f}us[{k}]" for v, (t, k) in res.items()) + flag, flush=True)
print(f"sum over the step: auto {tot_auto/1e3:.3f} ms, best-per-call {tot_best/1e3:.3f} ms ({100*(tot_auto/tot_best-1):.1f} % head-room)")

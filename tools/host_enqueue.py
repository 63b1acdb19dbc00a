"""Where does the HOST spend its time while enqueuing one step?  python tools/host_enqueue.py [config] [batch]
After a device sync (empty queue) one forward pass is enqueued and the host time of every libminddet_hip call (and of the whole pass) is
recorded WITHOUT synchronising; then the device time of the pass.  A call that blocks the host (hidden synchronisation, allocator) shows up
as a long host time."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet.models import Config, build_detector
from minddet_amd import _lib, nn_ops
from minddet_amd.data import synthetic_images

cfg_path = sys.argv[1] if len(sys.argv) > 1 else "configs/faster_rcnn/faster_rcnn_r50_fpn.py"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda:0")
cfg = Config.fromfile(cfg_path)
model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev)
H, W = cfg.data.input_hw
x = nn_ops.to_stem_layout(synthetic_images(B, H, W, seed=1, device=dev))
for _ in range(3):
    model.forward(x)
torch.cuda.synchronize()
host = collections.defaultdict(lambda: [0.0, 0])
orig_call = _lib.call

def timed_call(name, tensors, extra=None, stream=None):
    t0 = time.perf_counter()
    r = orig_call(name, tensors, extra=extra, stream=stream)
    d = host[name]
    d[0] += time.perf_counter() - t0; d[1] += 1
    return r

_lib.call = timed_call
for mod in (nn_ops,):
    pass
import minddet_amd.det_ops as det_ops
for rep in range(3):
    host.clear()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    model.forward(x)
    e1.record()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    tot_calls = sum(v[0] for v in host.values())
    print(f"rep {rep}: host enqueue of one pass {t_host*1e3:.2f} ms (inside library calls {tot_calls*1e3:.2f} ms, {sum(v[1] for v in host.values())} calls); "
          f"device {e0.elapsed_time(e1):.2f} ms; host+wait {t_all*1e3:.2f} ms", flush=True)
for k, v in sorted(host.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"   {k:28s} x{v[1]:3d}  host {v[0]*1e3:8.3f} ms  ({v[0]/v[1]*1e6:7.1f} us per call)")
# back-to-back passes: does the host run ahead of the device?
torch.cuda.synchronize()
t0 = time.perf_counter()
marks = []
for i in range(6):
    model.forward(x)
    marks.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("back-to-back: host time after pass i (ms):", [round(m * 1e3, 1) for m in marks], "all done", round((time.perf_counter() - t0) * 1e3, 1))

import os, sys
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import profile, ProfilerActivity
from minddet.models import Config, build_detector
from minddet_amd import nn_ops
from minddet_amd.data import synthetic_images
cfgp = sys.argv[1]
cfg = Config.fromfile(cfgp)
dev = torch.device("cuda:0")
model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev)
H, W = cfg.data.input_hw
x = nn_ops.to_stem_layout(synthetic_images(8, H, W, seed=1, device=dev))
for _ in range(2): model.forward(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.forward(x)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if "emcpy" in e.name or "copy_" in e.name.lower() or "aten::to" == e.name or "aten::_to_copy" == e.name or "aten::contiguous" == e.name or "aten::clone"==e.name]
seen = {}
for e in evs:
    st = [s for s in (e.stack or []) if "minddet" in s or "bench" in s]
    key = (e.name, tuple(st[:3]))
    seen[key] = seen.get(key, 0) + 1
for (n, st), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(c, n, " <- ".join(s.split("/")[-1] for s in st))

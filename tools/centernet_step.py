"""Times the CenterNet (ResNet-18 + DCN neck) inference step on synthetic 512x512 images: python tools/centernet_step.py [batch] [steps]."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minddet_amd import graphs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda:0"
m = graphs.CenterNet(depth=18, num_classes=80, seed=3).to(dev)
g = torch.Generator().manual_seed(0)
x = torch.zeros((B, 512, 512, 8))
x[..., :3] = torch.randn((B, 512, 512, 3), generator=g)
x = x.to(torch.bfloat16).to(dev)
for _ in range(5):
    m.forward(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    m.forward(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"CenterNet R18-DCN 512x512 batch {B}: {dt*1e3:.3f} ms/step, {B/dt:.1f} images/s")

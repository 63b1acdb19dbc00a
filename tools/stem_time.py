"""md_stem_pool timing (60 images), optionally with another build of the library (MD_LIB_OVERRIDE=path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet_amd import nn_ops, _lib
if os.environ.get('MD_LIB_OVERRIDE'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MD_LIB_OVERRIDE'])
from minddet_amd.data import synthetic_images
dev="cuda:0"
g=torch.Generator().manual_seed(0)
ps = nn_ops.pack_stem(torch.randn((64,3,7,7),generator=g)*0.1, bn=None, bias=torch.randn((64,),generator=g)*0.1).to(dev)
x = nn_ops.to_stem_layout(synthetic_images(60, 800, 1344, seed=1, device=dev))
y = nn_ops.stem_pool(x, ps); torch.cuda.synchronize()
ts=[]
for r in range(8):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): nn_ops.stem_pool(x, ps)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/5)
print("stem_pool b60: median %.4f ms min %.4f  checksum %.6f" % (sorted(ts)[4], min(ts), float(y.float().abs().mean())))

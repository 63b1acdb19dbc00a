"""A/B: the FPN lateral 1x1 convs with no / full-resolution / 2x-upsampled residual, auto kernel vs ping-pong forced."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from minddet_amd import nn_ops
B=60; dev="cuda:0"; g=torch.Generator().manual_seed(0)
for (H,W,Cin,Cout) in [(200,336,256,256),(100,168,512,256),(50,84,1024,256)]:
    w=torch.randn((Cout,Cin,1,1),generator=g)*(2.0/Cin)**0.5
    pc=nn_ops.pack_conv(w,bias=torch.zeros(Cout)).to(dev)
    x=torch.randn((B,H,W,Cin),generator=g).to(torch.bfloat16).to(dev)
    r_full=torch.randn((B,H,W,Cout),generator=g).to(torch.bfloat16).to(dev)
    r_up=torch.randn((B,(H+1)//2,(W+1)//2,Cout),generator=g).to(torch.bfloat16).to(dev)
    def t(f):
        for _ in range(2): f()
        ts=[]
        for _ in range(5):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): f()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/3)
        return sorted(ts)[2]*1e3
    a=t(lambda: nn_ops.conv2d(x,pc)); b=t(lambda: nn_ops.conv2d(x,pc,residual=r_full)); c=t(lambda: nn_ops.conv2d(x,pc,residual=r_up,res_upsample=True))
    a15=t(lambda: nn_ops.conv2d(x,pc,variant=15)); c15=t(lambda: nn_ops.conv2d(x,pc,residual=r_up,res_upsample=True,variant=15))
    print(f"   ping-pong forced: none {a15:.0f}us  res_up {c15:.0f}us")
    by=2.0*B*H*W*(Cin+Cout)
    print(f"{H}x{W}x{Cin}->{Cout}: none {a:.0f}us ({by/a/1e6:.2f}TB/s)  full-res {b:.0f}us ({(by+2.0*B*H*W*Cout)/b/1e6:.2f})  res_up {c:.0f}us ({(by+0.5*B*H*W*Cout)/c/1e6:.2f})")

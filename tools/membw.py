"""HBM microbenchmarks with torch kernels: fill (write only), sum (read only), copy (1:1), add (2 reads : 1 write)."""
import torch
n = 1100 * 1024 * 1024 // 2
x = torch.randn(n // 4, device="cuda").to(torch.bfloat16).repeat(4)[:n].contiguous()
y = torch.empty_like(x); z = torch.empty_like(x)
def t(f, bytes_, name):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:8s} {ms*1e3:8.1f} us  {bytes_/ms/1e9:6.2f} TB/s")
B = x.numel() * 2
t(lambda: y.fill_(1.0), B, "fill")
t(lambda: x.view(torch.int16).sum(), B, "sum")
t(lambda: y.copy_(x), 2 * B, "copy")
t(lambda: torch.add(x, y, out=z), 3 * B, "add")

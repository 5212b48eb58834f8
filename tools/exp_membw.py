"""Achievable write / copy / read bandwidth at the sizes of the expand layers' outputs (torch kernels, HIP events)."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (25, 50, 100, 176, 201, 400, 1000):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=dev); y = torch.empty_like(x)
    tw = t(lambda: x.zero_()); tc = t(lambda: y.copy_(x)); tr = t(lambda: x.view(torch.int16).max())
    print(f"{mb:5d} MB  fill {mb/1e3/tw:6.0f} GB/s ({tw*1e6:6.1f} us)  copy {2*mb/1e3/tc:6.0f} GB/s ({tc*1e6:6.1f} us)  read {mb/1e3/tr:6.0f} GB/s", flush=True)

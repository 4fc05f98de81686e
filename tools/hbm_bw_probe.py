#!/usr/bin/env python3
"""What plain streaming kernels (torch copy / fill / scale) reach on this GPU: the practical HBM ceiling the HBM-bound
products of the small towers are judged against (sizes up to ~230 MB live in the 256 MB Infinity Cache)."""
import torch
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for mb in (29, 58, 115, 230, 460, 1840):
    n = mb * 1000 * 1000 // 4
    a = torch.randn(n, device="cuda"); b = torch.empty_like(a)
    tc = timeit(lambda: b.copy_(a)); tf = timeit(lambda: b.fill_(1.0)); tr = timeit(lambda: a.sum())
    tm = timeit(lambda: torch.mul(a, 2.0, out=b))
    print(f"{mb:5d} MB: copy {tc:7.1f} us = {2*mb/tc:5.2f} TB/s (r+w)   fill {tf:7.1f} us = {mb/tf:5.2f} TB/s   sum {tr:7.1f} us = {mb/tr:5.2f} TB/s   mul {tm:7.1f} us = {2*mb/tm:5.2f} TB/s")

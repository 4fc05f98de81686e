#!/usr/bin/env python3
"""msn_attention_fwd / bwd at the reference's own tower shapes, per-GPU batch 1024, with key padding masks:
spectrum tower (220 tokens, e 32, 2 heads of 16) and light-curve tower (200 tokens, e 64, 8 heads of 8), for the
kernel families (1 = vector ALU, 2 = matrix cores)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_supernovae_amd import ops, _lib


def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


L = _lib.lib()
for name, B, T, E, H in [("spectrum", 1024, 220, 32, 2), ("lightcurve", 1024, 200, 64, 8), ("spectrum 1024 bins", 64, 1024, 32, 2),
                         ("vit-s", 1024, 65, 384, 6), ("reference default e256 / 2 heads", 1024, 200, 256, 2),
                         ("128-wide heads, 100 tokens", 1024, 100, 256, 2)]:
    torch.manual_seed(0)
    qkv = torch.randn(B, T, 3 * E, device="cuda")
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    lens = torch.randint(T // 2, T + 1, (B,), device="cuda")
    mask = None if name == "vit-s" else torch.arange(T, device="cuda")[None, :] < lens[:, None]
    dout = torch.randn(B, T, E, device="cuda"); dqkv = torch.empty_like(qkv)
    scale = 1 / math.sqrt(E)
    ref = None
    for path in (1, 2):
        _lib.check(L.msn_set_attention_path(path))
        out, lse = ops.attention_fwd(q, k, v, mask, H, scale)
        run_b = lambda: ops.attention_bwd(q, k, v, mask, H, scale, out, lse, dout, dqkv[..., :E], dqkv[..., E:2 * E], dqkv[..., 2 * E:])
        run_b()
        if ref is None: ref = (out.clone(), dqkv.clone())
        err = max((out - ref[0]).abs().max().item(), (dqkv - ref[1]).abs().max().item())
        tf = timeit(lambda: ops.attention_fwd(q, k, v, mask, H, scale)); tb = timeit(run_b)
        print(f"{name:20s} path {path}: fwd {tf:7.1f} us  bwd {tb:7.1f} us   max |diff to path 1| {err:.2e}", flush=True)
L.msn_set_attention_path(0)

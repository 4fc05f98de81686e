#!/usr/bin/env python3
"""msn_attention_fwd / bwd on the reference's spectrum tower at 1024 bins (e 32, 2 heads of 16): vector-ALU (1) against
the chunked matrix-core kernels (2)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_supernovae_amd import ops, _lib
def timeit(fn, iters=20, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for B, T, E, H in [(1024, 1024, 32, 2), (256, 1024, 384, 6)]:
    qkv = torch.randn(B, T, 3 * E, device="cuda")
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    dout = torch.randn(B, T, E, device="cuda"); dqkv = torch.empty_like(qkv)
    scale = 1 / math.sqrt(E)
    for path in (1, 2):
        _lib.check(_lib.lib().msn_set_attention_path(path))
        out, lse = ops.attention_fwd(q, k, v, None, H, scale)
        tf = timeit(lambda: ops.attention_fwd(q, k, v, None, H, scale))
        tb = timeit(lambda: ops.attention_bwd(q, k, v, None, H, scale, out, lse, dout, dqkv[..., :E], dqkv[..., E:2 * E], dqkv[..., 2 * E:]))
        pairs = B * H * T * T
        print(f"B={B} T={T} e={E} h={H} path {path}: fwd {tf:8.1f} us ({tf * 1e-6 * 2.4e9 * 1024 / (pairs / 64):.0f} cyc / 64 pairs / SIMD)   bwd {tb:8.1f} us")
_lib.lib().msn_set_attention_path(0)

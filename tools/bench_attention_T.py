#!/usr/bin/env python3
"""msn_attention_fwd / bwd of the ViT-S tower (e 384, 6 heads) at equal bytes over sequence lengths around whole tiles:
16n (no ragged token), 16n + 1 (ragged token on the vector ALU), others (zero-padded tiles)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_supernovae_amd import ops, _lib
def timeit(fn, iters=50, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
E, H = 384, 6
for T in (48, 64, 65, 80, 96, 128):
    B = 1024 * 65 // T
    qkv = torch.randn(B, T, 3 * E, device="cuda")
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    dout = torch.randn(B, T, E, device="cuda"); dqkv = torch.empty_like(qkv)
    scale = 1 / math.sqrt(E)
    out, lse = ops.attention_fwd(q, k, v, None, H, scale)
    tf = timeit(lambda: ops.attention_fwd(q, k, v, None, H, scale))
    tb = timeit(lambda: ops.attention_bwd(q, k, v, None, H, scale, out, lse, dout, dqkv[..., :E], dqkv[..., E:2 * E], dqkv[..., 2 * E:]))
    nt = (T + 15) // 16
    print(f"T={T:4d} B={B} tiles^2*B*H={nt*nt*B*H/1e3:.0f}k: fwd {tf:8.1f} us ({tf/(nt*nt*B*H)*1e3:.3f} ns/tilepair)  bwd {tb:8.1f} us ({tb/(nt*nt*B*H)*1e3:.3f})  bytes-fwd {B*T*E*16/1e6:.0f} MB")

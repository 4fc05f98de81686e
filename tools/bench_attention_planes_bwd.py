#!/usr/bin/env python3
"""Backward of the plane attention at the spectrum-tower shape, one mode per process (for rocprofv3 / tools/pmc_kernels.sh):
    python tools/bench_attention_planes_bwd.py <msn_set_attention_planes mode: 1 | 3 | 5> <launches> [tokens]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_supernovae_amd import ops, _lib
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, T, E, H = 1024, int(sys.argv[3]) if len(sys.argv) > 3 else 1024, 32, 2
qkv = torch.randn(B, T, 3 * E, device="cuda")
q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
dout = torch.randn(B, T, E, device="cuda"); dqkv = torch.empty_like(qkv)
scale = 1 / math.sqrt(E)
out, lse = ops.attention_fwd(q, k, v, None, H, scale)
_lib.check(_lib.lib().msn_set_attention_planes(mode))
fn = lambda: ops.attention_bwd(q, k, v, None, H, scale, out, lse, dout, dqkv[..., :E], dqkv[..., E:2 * E], dqkv[..., 2 * E:])
for _ in range(iters): fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters): fn()
e.record(); torch.cuda.synchronize()
print(f"T={T} mode {mode}: bwd {s.elapsed_time(e) / iters * 1e3:8.1f} us", flush=True)

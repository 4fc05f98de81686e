#!/usr/bin/env python3
"""Long-sequence attention of the reference-native towers: exact-fp32 matrix-core kernels (planes 0), the fp32-grade bf16
plane kernels (msn_set_attention_planes 1 = default: one-pass backward at these batches; --variants: 3 = the two-kernel backward),
and for 8-wide heads the vector-ALU kernels.
Per launch: us, and useful TFLOP/s (4 T^2 hd flop per (b, h) forward, 10 T^2 hd backward)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_supernovae_amd import ops, _lib
def timeit(fn, iters=10, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
SHAPES = [(1024, 1024, 32, 2, "spectrum tower, 1024 bins"), (1024, 220, 32, 2, "Maven spectrum tower"),
          (1024, 200, 64, 8, "light-curve tower (8-wide heads)")]
full = "--variants" in sys.argv
for B, T, E, H, what in SHAPES:
    hd = E // H
    qkv = torch.randn(B, T, 3 * E, device="cuda")
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    dout = torch.randn(B, T, E, device="cuda"); dqkv = torch.empty_like(qkv)
    scale = 1 / math.sqrt(E)
    variants = [("fp32 MFMA", 2, 0), ("planes (default)", 2, 1)]
    if full:
        variants += [("planes, two-kernel bwd", 2, 3)]
    if hd < 16:
        variants.insert(0, ("vector ALU", 1, 0))
    for name, path, planes in variants:
        _lib.check(_lib.lib().msn_set_attention_path(path))
        ops.set_attention_planes(1 if planes else 0)
        out, lse = ops.attention_fwd(q, k, v, None, H, scale)            # valid statistics for the backward whatever the variant
        ops.set_attention_planes(planes)
        tf = timeit(lambda: ops.attention_fwd(q, k, v, None, H, scale))
        tb = timeit(lambda: ops.attention_bwd(q, k, v, None, H, scale, out, lse, dout, dqkv[..., :E], dqkv[..., E:2 * E], dqkv[..., 2 * E:]))
        fl = B * H * T * T * hd
        print(f"{what:34s} B={B} T={T} e={E} h={H} {name:22s}: fwd {tf:8.1f} us ({4 * fl / tf * 1e-6:6.1f} TFLOP/s)   bwd {tb:8.1f} us ({10 * fl / tb * 1e-6:6.1f} TFLOP/s)", flush=True)
_lib.lib().msn_set_attention_path(0)
ops.set_attention_planes(1)

#!/usr/bin/env python3
"""Micro-benchmark of the LayerNorm kernels on the ViT-S token matrix (GPU box only)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_supernovae_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for rows, cols in [(66560, 384), (204800, 64)]:
    x = torch.randn(rows, cols, device="cuda")
    g, b = torch.randn(cols, device="cuda"), torch.randn(cols, device="cuda")
    dy, add = torch.randn_like(x), torch.randn_like(x)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
    tf = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5))
    tb = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, add=add))
    nb = rows * cols * 4
    print(f"layernorm {rows}x{cols}: fwd {tf:7.1f} us ({2 * nb / tf / 1e6:5.2f} TB/s)   bwd+add {tb:7.1f} us ({4 * nb / tb / 1e6:5.2f} TB/s)")

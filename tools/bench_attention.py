#!/usr/bin/env python3
"""Micro-benchmark of msn_attention_fwd / bwd on the two towers' shapes, vector-ALU (1) vs matrix-core (2) path."""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_supernovae_amd import ops, _lib


def timeit(fn, iters=10, warm=2):
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


for tag, B, T, E, H, masked in [("lc  T=200 e64 h8", 1024, 200, 64, 8, True), ("vit T=65 e384 h6", 1024, 65, 384, 6, False)]:
    qkv = torch.randn(B, T, 3 * E, device="cuda")
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    mask = (torch.rand(B, T, device="cuda") > 0.2).to(torch.uint8) if masked else None
    dout = torch.randn(B, T, E, device="cuda")
    dqkv = torch.empty_like(qkv)
    scale = 1 / math.sqrt(E)
    for path in (1, 2):
        _lib.check(_lib.lib().msn_set_attention_path(path))
        out, lse = ops.attention_fwd(q, k, v, mask, H, scale)
        tf = timeit(lambda: ops.attention_fwd(q, k, v, mask, H, scale))
        tb = timeit(lambda: ops.attention_bwd(q, k, v, mask, H, scale, out, lse, dout, dqkv[..., :E], dqkv[..., E:2 * E], dqkv[..., 2 * E:]))
        print(f"{tag} path {path}: fwd {tf:8.1f} us   bwd {tb:8.1f} us")
_lib.lib().msn_set_attention_path(0)

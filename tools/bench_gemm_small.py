#!/usr/bin/env python3
"""fp32 GEMM plans on the headline's shapes at small per-GPU batches (strong scaling: 128 / 256 rows per GPU):
planned launch (128-wide tiles + tail split + finish pass) against 64-wide tiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_supernovae_amd import ops
from multimodal_supernovae_amd._lib import check, lib


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for B in (128, 256, 512):
    M = B * 65
    for N, K, ob, tag in [(384, 384, 1, "proj fwd"), (1152, 384, 1, "qkv fwd"), (1536, 384, 1, "ff1 fwd"), (384, 1536, 1, "ff2 fwd"),
                          (384, 1152, 0, "dqkv dgrad"), (384, 1536, 0, "ff1 dgrad"), (1536, 384, 0, "ff2 dgrad"), (384, 384, 0, "proj dgrad")]:
        a = torch.randn(M, K, device="cuda")
        b = torch.randn((K, N) if ob == 0 else (N, K), device="cuda")
        out = torch.empty(M, N, device="cuda")
        res = []
        for bn in (0, 64, 128):
            check(lib().msn_set_gemm_tile_n(bn))
            res.append(timeit(lambda: ops.sgemm(a, b, 0, ob, out=out)))
        check(lib().msn_set_gemm_tile_n(0))
        print(f"B={B:4d} {tag:11s} M={M:6d} N={N:5d} K={K:5d}  planned {res[0]:7.1f} us  bn=64 {res[1]:7.1f} us  bn=128 {res[2]:7.1f} us  "
              f"({2.0 * M * N * K / res[0] / 1e6:6.1f} / {2.0 * M * N * K / res[1] / 1e6:6.1f} TFLOP/s)", flush=True)

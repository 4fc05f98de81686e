#!/usr/bin/env python3
"""Work-list (stream-K) launches against the planned flat launches on the headline's ViT-S/8 shapes:
forward / dgrad products one at a time (msn_set_gemm_list 2 / 3) and the backward pair of every Linear (ops.dgrad_wgrad
against msn_sgemm + msn_wgrad_bias)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_supernovae_amd import ops


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


batches = [int(v) for v in sys.argv[1:]] or [128, 256, 512, 1024]
for B in batches:
    M = B * 65
    tot = [0.0, 0.0]
    for N, K, ob, tag in [(384, 384, 1, "proj fwd"), (1152, 384, 1, "qkv fwd"), (1536, 384, 1, "ff1 fwd"), (384, 1536, 1, "ff2 fwd")]:
        a = torch.randn(M, K, device="cuda")
        b = torch.randn((K, N) if ob == 0 else (N, K), device="cuda")
        out = torch.empty(M, N, device="cuda")
        ops.set_gemm_list(2)
        t0 = timeit(lambda: ops.sgemm(a, b, 0, ob, out=out))
        ops.set_gemm_list(3)
        t1 = timeit(lambda: ops.sgemm(a, b, 0, ob, out=out))
        ops.set_gemm_list(1)
        tot[0] += t0
        tot[1] += t1
        print(f"B={B:4d} {tag:10s} M={M:6d} N={N:5d} K={K:5d}  flat {t0:7.1f} us ({2.0 * M * N * K / t0 / 1e6:6.1f} TF)  "
              f"work-list {t1:7.1f} us ({2.0 * M * N * K / t1 / 1e6:6.1f} TF)", flush=True)
    for n_in, n_out, tag in [(384, 384, "proj bwd"), (384, 1152, "qkv bwd"), (384, 1536, "ff1 bwd"), (1536, 384, "ff2 bwd")]:
        dy = torch.randn(M, n_out, device="cuda")
        w = torch.randn(n_out, n_in, device="cuda")
        x = torch.randn(M, n_in, device="cuda")
        ops.set_gemm_list(False)
        t0 = timeit(lambda: ops.dgrad_wgrad(dy, w, x))
        ops.set_gemm_list(True)
        t1 = timeit(lambda: ops.dgrad_wgrad(dy, w, x))
        fl = 4.0 * M * n_in * n_out
        tot[0] += t0
        tot[1] += t1
        print(f"B={B:4d} {tag:10s} rows={M:6d} in={n_in:5d} out={n_out:5d}  separate {t0:7.1f} us ({fl / t0 / 1e6:6.1f} TF)  "
              f"one launch {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TF)", flush=True)
    print(f"B={B:4d} sum over the 4 forward products + 4 backward pairs of a block: {tot[0]:8.1f} -> {tot[1]:8.1f} us", flush=True)

#!/usr/bin/env python3
"""Micro-benchmark of msn_sgemm on the shapes of the headline workload (GPU box only).
usage: bench_gemm.py [precision ...]   (0 = fp32 MFMA, 1 = split-bf16 x3, 2 = bf16)"""
import sys
import os
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
    return s.elapsed_time(e) / iters * 1e-3

shapes = [  # (M, N, K, opA, opB, tag)
    (66560, 384, 384, 0, 1, "vit-s proj fwd"), (66560, 1152, 384, 0, 1, "vit-s qkv fwd"),
    (66560, 1536, 384, 0, 1, "vit-s ff1 fwd"), (66560, 384, 1536, 0, 1, "vit-s ff2 fwd"),
    (66560, 384, 1536, 0, 0, "vit-s ff1 dgrad"), (1536, 384, 66560, 1, 0, "vit-s ff1 wgrad"),
    (384, 1536, 66560, 1, 0, "vit-s ff2 wgrad"),
    (51200, 64, 64, 0, 1, "lc e64 proj"), (51200, 256, 64, 0, 1, "lc ff1"), (51200, 64, 256, 0, 1, "lc ff2"),
    (4096, 4096, 4096, 0, 1, "4096^3 NT"), (4096, 4096, 4096, 0, 0, "4096^3 NN"), (4096, 4096, 4096, 1, 0, "4096^3 TN"),
]
precs = [int(v) for v in sys.argv[1:]] or [0]
if os.environ.get("MSN_GEMM_VARIANT"):
    ops.set_gemm_variant(int(os.environ["MSN_GEMM_VARIANT"]))
for prec in precs:
    print(f"--- precision {prec}")
    for M, N, K, oa, ob, tag in shapes:
        a = torch.randn((M, K) if oa == 0 else (K, M), device="cuda")
        b = torch.randn((K, N) if ob == 0 else (N, K), device="cuda")
        out = torch.empty(M, N, device="cuda")
        t = timeit(lambda: ops.sgemm(a, b, oa, ob, out=out, precision=prec))
        print(f"{tag:18s} M={M:6d} N={N:5d} K={K:6d}  {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:7.1f} TFLOP/s  "
              f"{(M*K+K*N+M*N)*4/t/1e9:8.0f} GB/s", flush=True)
        if tag == "vit-s ff1 fwd":      # the same product with its training epilogue: + bias, GELU, gelu' saved
            bias, aux = torch.randn(N, device="cuda"), torch.empty(M, N, device="cuda")
            t = timeit(lambda: ops.sgemm(a, b, oa, ob, out=out, precision=prec, bias=bias, epilogue=ops.EPI_GELU, aux=aux))
            print(f"{'  + bias/GELU/aux':18s} M={M:6d} N={N:5d} K={K:6d}  {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:7.1f} TFLOP/s", flush=True)

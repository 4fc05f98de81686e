#!/usr/bin/env python3
"""Does the fp32 GEMM rate depend on the operand DATA (matrix-pipe power management) rather than on the kernel?
The same launch on all-zero, constant, small-integer and N(0,1) operands."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_supernovae_amd import ops


def timeit(fn, iters=30, warm=10):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for (M, N, K, oa, ob) in [(4096, 4096, 4096, 0, 1), (66560, 1536, 384, 0, 1), (1536, 384, 66560, 1, 0)]:
    sa, sb = ((M, K) if oa == 0 else (K, M)), ((K, N) if ob == 0 else (N, K))
    out = torch.empty(M, N, device="cuda")
    for tag, mk in [("zeros", lambda s: torch.zeros(s, device="cuda")), ("ones", lambda s: torch.ones(s, device="cuda")),
                    ("ints 0..3", lambda s: torch.randint(0, 4, s, device="cuda").float()),
                    ("N(0,1)", lambda s: torch.randn(s, device="cuda")),
                    ("N(0,1) bf16-rounded", lambda s: torch.randn(s, device="cuda").bfloat16().float())]:
        a, b = mk(sa), mk(sb)
        us = timeit(lambda: ops.sgemm(a, b, oa, ob, out=out))
        print(f"M={M:6d} N={N:5d} K={K:6d} op={oa}{ob} {tag:20s} {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s", flush=True)

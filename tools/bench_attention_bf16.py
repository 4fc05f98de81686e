#!/usr/bin/env python3
"""The bf16 attention kernels of the cfg5 ViT-B/16 (msn_attention_bf16_fwd / _bwd): us per launch, HIP events.
    python tools/bench_attention_bf16.py [B T heads]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_supernovae_amd import ops  # noqa: E402

B, T, heads = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 197, 12)
e = heads * 64
qkv = (torch.randn(B * T, 3 * e, device="cuda") * 0.5).to(torch.bfloat16)
da = (torch.randn(B * T, e, device="cuda") * 0.5).to(torch.bfloat16)
scale = 1.0 / math.sqrt(64)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ab, lse = ops.attention_bf16_fwd(qkv, B, T, heads, scale)
tf = timeit(lambda: ops.attention_bf16_fwd(qkv, B, T, heads, scale))
tb = timeit(lambda: ops.attention_bf16_bwd(qkv, ab, da, lse, B, T, heads, scale, want_colsum=True))
fl = B * heads * T * T * 64
print(f"bf16 attention B={B} T={T} {heads} x 64: fwd {tf:7.1f} us ({4 * fl / tf * 1e-6:6.1f} TFLOP/s, {(4 * B * T * e * 2) / tf * 1e-6:5.2f} TB/s)   "
      f"bwd {tb:7.1f} us ({10 * fl / tb * 1e-6:6.1f} TFLOP/s)")

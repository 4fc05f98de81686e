#!/usr/bin/env python3
"""Forward of the ViT towers' self-attention with the plane output (msn_attention_fwd_planes): us per launch, HIP events.
    python tools/bench_attention_fwd_planes.py [B T heads hd]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_supernovae_amd import ops  # noqa: E402

B, T, heads, hd = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (1024, 65, 6, 64)
E = heads * hd
qkv = torch.randn(B, T, 3 * E, device="cuda")
scale = 1.0 / math.sqrt(hd)
fn = lambda: ops.attention_fwd_planes(qkv, heads, scale)
for _ in range(5):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    fn()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 30 * 1e3
print(f"attention forward + plane output, B={B} T={T} {heads} x {hd}: {t:7.1f} us  ({4.0 * B * heads * T * T * hd / t * 1e-6:6.1f} TFLOP/s)")

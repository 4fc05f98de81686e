#!/usr/bin/env python3
"""Run one msn_sgemm shape a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_supernovae_amd import ops
M, N, K, oa, ob = [int(v) for v in sys.argv[1:6]]
a = torch.randn((M, K) if oa == 0 else (K, M), device="cuda")
b = torch.randn((K, N) if ob == 0 else (N, K), device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(5):
    ops.sgemm(a, b, oa, ob, out=out)
torch.cuda.synchronize()

#!/usr/bin/env python3
"""Light-curve tower products (204 800 token rows, K and N of 64 .. 256): native fp32 kernel against plane split + plane GEMM."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multimodal_supernovae_amd import ops


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


M = 1024 * 200
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K in [("qkv", 192, 64), ("unify", 64, 64), ("ff1", 256, 64), ("ff2", 64, 256), ("dqkv", 64, 192)]:
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.1
    t0 = timed(lambda: ops.sgemm(a, w, ops.OP_N, ops.OP_T, precision=ops.PREC_F32))
    ts = timed(lambda: ops.plane_split(a, 3))
    ap, wp = ops.plane_split(a, 3), ops.plane_split(w, 3)
    tg = timed(lambda: ops.pgemm_nt(ap, wp))
    ref = a.double() @ w.double().t()
    e0 = (ops.sgemm(a, w, ops.OP_N, ops.OP_T, precision=ops.PREC_F32).double() - ref).abs().max().item()
    e1 = (ops.pgemm_nt(ap, wp).double() - ref).abs().max().item()
    print(f"{name:6s} N={N:4d} K={K:4d}: native fp32 {t0:6.1f} us | split {ts:5.1f} + plane GEMM {tg:6.1f} us | max err native {e0:.2e} planes {e1:.2e}", flush=True)

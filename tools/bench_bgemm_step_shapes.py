#!/usr/bin/env python3
"""Every bf16-resident product of one cfg5 ViT-B/16 block (M = 512 x 197 rows) with the epilogue the trunk gives it: us per launch and
TFLOP/s, HIP events (functional._Bf16VitTrunk: forward qkv / proj / fc1 / fc2, backward dgrads and wgrads)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_supernovae_amd import ops  # noqa: E402


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


M, e = int(sys.argv[1]) if len(sys.argv) > 1 else 512 * 197, 768
g = torch.Generator(device="cuda").manual_seed(0)
rb = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.3).to(torch.bfloat16)
rf = lambda *s: torch.randn(*s, device="cuda", generator=g)
x, x3, x4 = rb(M, e), rb(M, 3 * e), rb(M, 4 * e)
res = rf(M, e)
tot_t = tot_f = 0.0
w = {k: rb(*k) for k in [(3 * e, e), (e, e), (4 * e, e), (e, 4 * e), (e, 3 * e)]}
b3, b1, b4 = rf(3 * e), rf(e), rf(4 * e)
fixed = [
    ("fwd qkv   (bias, bf16 out)        ", 3 * e, e, lambda: ops.bgemm_nt(x, w[(3 * e, e)], bias=b3, out_bf16=True)),
    ("fwd proj  (bias + residual, fp32) ", e, e, lambda: ops.bgemm_nt(x, w[(e, e)], bias=b1, epilogue=ops.BEPI_ADD, aux=res)),
    ("fwd fc1   (bias, GELU, 2 x bf16)  ", 4 * e, e, lambda: ops.bgemm_nt(x, w[(4 * e, e)], bias=b4, epilogue=ops.BEPI_GELU, out_bf16=True)),
    ("fwd fc2   (bias + residual, fp32) ", e, 4 * e, lambda: ops.bgemm_nt(x4, w[(e, 4 * e)], bias=b1, epilogue=ops.BEPI_ADD, aux=res)),
    ("bwd dpre  (x GELU', colsum, bf16) ", 4 * e, e, lambda: ops.bgemm_nt(x, w[(4 * e, e)], epilogue=ops.BEPI_GELU_BWD, aux=x4, out_bf16=True, want_colsum=True)),
    ("bwd dh2   (bf16 out, K = 3072)    ", e, 4 * e, lambda: ops.bgemm_nt(x4, w[(e, 4 * e)], out_bf16=True)),
    ("bwd da    (bf16 out, K = 768)     ", e, e, lambda: ops.bgemm_nt(x, w[(e, e)], out_bf16=True)),
    ("bwd dh1   (bf16 out, K = 2304)    ", e, 3 * e, lambda: ops.bgemm_nt(x3, w[(e, 3 * e)], out_bf16=True)),
    ("bwd dW2   (TN 768 x 3072)         ", e, 4 * e, lambda: ops.bgemm_tn(x, x4)),
    ("bwd dW1   (TN 3072 x 768)         ", 4 * e, e, lambda: ops.bgemm_tn(x4, x)),
    ("bwd dWo   (TN 768 x 768)          ", e, e, lambda: ops.bgemm_tn(x, x)),
    ("bwd dWqkv (TN 2304 x 768)         ", 3 * e, e, lambda: ops.bgemm_tn(x3, x)),
]
for name, N, K, fn in fixed:
    t = timed(fn)
    fl = 2.0 * M * N * K
    tot_t += t
    tot_f += fl
    print(f"{name}: {t:7.1f} us  {fl / t * 1e-6:7.1f} TFLOP/s", flush=True)
print(f"block total: {tot_t:7.1f} us  {tot_f / tot_t * 1e-6:7.1f} TFLOP/s = {tot_f / tot_t * 1e-6 / 2500:.3f} of the bf16 peak")

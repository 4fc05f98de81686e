#!/usr/bin/env python3
"""The epilogues of the bf16-resident NT product on the cfg5 feed-forward shapes (M = 512 x 197 rows): plain bf16 output,
GELU forward (activation + saved pre-activation), GELU backward (x gelu'(saved pre-activation)).  us per launch, HIP events."""
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


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 512 * 197
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    x, w1, w2t = rnd(M, 768), rnd(3072, 768), rnd(3072, 768)
    b1 = torch.randn(3072, device="cuda", generator=g)
    dy = rnd(M, 768)
    pre = rnd(M, 3072)
    fl = 2.0 * M * 768 * 3072
    rows = [("fc1 fwd, plain bf16 out", lambda: ops.bgemm_nt(x, w1, b1, out_bf16=True)),
            ("fc1 fwd, fp32 out", lambda: ops.bgemm_nt(x, w1, b1)),
            ("fc1 fwd, GELU (+ saved pre-activation)", lambda: ops.bgemm_nt(x, w1, b1, epilogue=ops.BEPI_GELU, out_bf16=True)),
            ("fc2 dgrad, plain bf16 out", lambda: ops.bgemm_nt(dy, w2t, out_bf16=True)),
            ("fc2 dgrad, x gelu'(pre)", lambda: ops.bgemm_nt(dy, w2t, epilogue=ops.BEPI_GELU_BWD, aux=pre, out_bf16=True))]
    for name, fn in rows:
        t = timed(fn)
        print(f"{name:42s}: {t:7.1f} us  {fl / t * 1e-6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()

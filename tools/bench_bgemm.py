#!/usr/bin/env python3
"""bf16-resident GEMM throughput on the BASELINE cfg5 (ViT-B/16) shapes, random operands, HIP events, warm clocks."""
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
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 512 * 197
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    for name, N, K in [("qkv   fwd", 2304, 768), ("proj  fwd", 768, 768), ("fc1   fwd", 3072, 768), ("fc2   fwd", 768, 3072),
                       ("dqkv dgrad", 768, 2304), ("dfc1 dgrad", 768, 3072)]:
        a, w = rnd(M, K), rnd(N, K)
        t = timed(lambda: ops.bgemm_nt(a, w))
        tb = timed(lambda: ops.bgemm_nt(a, w, out_bf16=True))
        print(f"NT {name}: M={M} N={N:5d} K={K:5d}  fp32 out {2.0 * M * N * K / t / 1e12:7.1f} TFLOP/s   bf16 out "
              f"{2.0 * M * N * K / tb / 1e12:7.1f} TFLOP/s", flush=True)
    for name, N, K in [("wqkv wgrad", 2304, 768), ("wo   wgrad", 768, 768), ("w1   wgrad", 3072, 768), ("w2   wgrad", 768, 3072)]:
        dy, x = rnd(M, N), rnd(M, K)
        t = timed(lambda: ops.bgemm_tn(dy, x))
        print(f"TN {name}: M={M} N={N:5d} K={K:5d}  {2.0 * M * N * K / t / 1e12:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()

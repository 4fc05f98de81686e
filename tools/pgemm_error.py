#!/usr/bin/env python3
"""Error of the plane GEMMs against fp64, next to the native fp32 MFMA kernel's, as a function of the reduction length
(N(0,1) operands): max and RMS absolute error and their ratio plane / native."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_supernovae_amd import ops  # noqa: E402


def stats(c, ref):
    e = (c.double() - ref).abs()
    return float(e.max()), float(e.pow(2).mean().sqrt())


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    M, N = 2080, 384
    print("NT  (M = 2080, N = 384)")
    for K in (192, 384, 768, 1536, 3072, 6144):
        a, w = torch.randn(M, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g)
        ref = a.double() @ w.double().T
        nat = stats(ops.sgemm(a, w, ops.OP_N, ops.OP_T, precision=ops.PREC_F32), ref)
        line = f"  K={K:5d} native max {nat[0]:.3e} rms {nat[1]:.3e}"
        for pl in (3, 2):
            s = stats(ops.pgemm_nt(ops.plane_split(a, pl), ops.plane_split(w, pl)), ref)
            line += f" | {pl}pl max {s[0]:.3e} ({s[0] / nat[0]:5.2f}x) rms {s[1]:.3e} ({s[1] / nat[1]:5.2f}x)"
        print(line, flush=True)
    print("TN  (N = 384, K = 384; reduction length R)")
    for R in (2048, 8320, 33280, 66560):
        dy, x = torch.randn(R, 384, device="cuda", generator=g), torch.randn(R, 384, device="cuda", generator=g)
        ref = dy.double().T @ x.double()
        nat = stats(ops.sgemm(dy, x, ops.OP_T, ops.OP_N, precision=ops.PREC_F32), ref)
        line = f"  R={R:6d} native max {nat[0]:.3e} rms {nat[1]:.3e}"
        for pl in (3, 2):
            s = stats(ops.pgemm_tn(ops.plane_split(dy, pl), ops.plane_split(x, pl)), ref)
            line += f" | {pl}pl max {s[0]:.3e} ({s[0] / nat[0]:5.2f}x) rms {s[1]:.3e} ({s[1] / nat[1]:5.2f}x)"
        print(line, flush=True)


if __name__ == "__main__":
    main()

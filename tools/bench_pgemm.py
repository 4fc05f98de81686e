#!/usr/bin/env python3
"""Plane-GEMM throughput on the headline (ViT-S/8, 1024 cutouts x 65 tokens) shapes, random operands, HIP events, warm
clocks; fp32-equivalent TFLOP/s (2 M N K / t) next to the native fp32 MFMA kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_supernovae_amd import ops  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024 * 65
    planes_list = [int(v) for v in os.environ.get("PLANES", "3,2").split(",")]
    if os.environ.get("TAIL") == "0":           # the tiles of the last, partly filled round multiplied whole
        ops.set_pgemm_tail_split(False)
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g) * 0.3
    shapes = [("qkv   fwd", 1152, 384), ("proj  fwd", 384, 384), ("fc1   fwd", 1536, 384), ("fc2   fwd", 384, 1536),
              ("dqkv dgrad", 384, 1152), ("patch fwd", 384, 192)]
    for name, N, K in shapes:
        a, w = rnd(M, K), rnd(N, K)
        fl = 2.0 * M * N * K
        t0 = timed(lambda: ops.sgemm(a, w, ops.OP_N, ops.OP_T, precision=ops.PREC_F32))
        line = f"NT {name}: M={M} N={N:5d} K={K:5d}  native fp32 {fl / t0 / 1e12:6.1f}"
        for pl in planes_list:
            ap, wp = ops.plane_split(a, pl), ops.plane_split(w, pl)
            ts = timed(lambda: ops.plane_split(a, pl))
            if pl == 3:
                t = timed(lambda: ops.pgemm_nt(ap, wp))
                line += f" | 3pl {fl / t / 1e12:6.1f} ({t * 1e6:6.0f} us)"
            else:
                t = timed(lambda: ops.pgemm_nt(ap, wp))
                line += f" | {'f16 2' if pl == ops.F16_PLANES else pl}pl {fl / t / 1e12:6.1f} ({t * 1e6:6.0f} us)"
            if pl == ops.F16_PLANES:              # fp32 results only; the split includes the pass for the largest magnitude
                bias, dact = rnd(N), rnd(M, N)
                tg = timed(lambda: ops.pgemm_nt(ap, wp, bias=bias, epilogue=ops.EPI_GELU, aux=True))
                tb = timed(lambda: ops.pgemm_nt(ap, wp, epilogue=ops.EPI_GELU_BWD, aux=dact, want_colsum=True))
                line += f" gelu {tg * 1e6:5.0f} us gelu' {tb * 1e6:5.0f} us; split {ts * 1e6:5.0f} us"
                continue
            tp = timed(lambda: ops.pgemm_nt(ap, wp, out_planes=True))
            line += f" planes-out {fl / tp / 1e12:6.1f} ({tp * 1e6:5.0f} us)"
            if N % 16 == 0 and os.environ.get("GELU"):
                bias = rnd(N)
                tg = timed(lambda: ops.pgemm_nt(ap, wp, bias=bias, epilogue=ops.EPI_GELU, aux=True, out_planes=True))
                dact = rnd(M, N)
                tb = timed(lambda: ops.pgemm_nt(ap, wp, epilogue=ops.EPI_GELU_BWD, aux=dact, out_planes=True, want_colsum=True))
                line += f" gelu {tg * 1e6:5.0f} us gelu' {tb * 1e6:5.0f} us"
            line += f"; split {ts * 1e6:5.0f} us"
        print(line, flush=True)
    if os.environ.get("NT_ONLY"):
        return
    for name, N, K in [("wqkv wgrad", 1152, 384), ("wo   wgrad", 384, 384), ("w1   wgrad", 1536, 384), ("w2   wgrad", 384, 1536)]:
        dy, x = rnd(M, N), rnd(M, K)
        fl = 2.0 * M * N * K
        t0 = timed(lambda: ops.wgrad_bias(dy, x, precision=ops.PREC_F32))
        line = f"TN {name}: M={M} N={N:5d} K={K:5d}  native fp32 {fl / t0 / 1e12:6.1f}"
        for pl in planes_list:
            dp, xp = ops.plane_split(dy, pl), ops.plane_split(x, pl)
            t = timed(lambda: ops.pgemm_tn(dp, xp))
            line += f" | {'f16 2' if pl == ops.F16_PLANES else pl}pl {fl / t / 1e12:6.1f} ({t * 1e6:6.0f} us)"
        print(line, flush=True)


if __name__ == "__main__":
    main()

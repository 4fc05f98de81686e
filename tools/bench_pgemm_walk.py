#!/usr/bin/env python3
"""Tile walk of the plane NT kernel (msn_set_pgemm_walk: column groups x super-rows) on the headline shapes: us per launch, HIP
events, warm clocks, configurations interleaved in one process (boxes differ by up to 7 %).

    python tools/bench_pgemm_walk.py                       # sweep, every shape
    python tools/bench_pgemm_walk.py --one fc1g 6 3 40     # ONE configuration, 40 launches (for rocprofv3 --pmc FETCH_SIZE passes)
MSN_HIP_LIB=tools/microbench/ablate/libmsn_PG_ANT.so runs the same sweep on the build whose A pieces carry the nt cache policy."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_supernovae_amd import ops  # noqa: E402

M = 1024 * 65
# name: (N, K, form)   form: f = fp32 output, p = plane output, g = plane output + GELU (+ saved derivative), b = GELU' backward + column sums
SHAPES = {"qkv": (1152, 384, "f"), "proj": (384, 384, "f"), "fc1p": (1536, 384, "p"), "fc1g": (1536, 384, "g"), "fc2": (384, 1536, "f"),
          "dfc2b": (1536, 384, "b"), "dqkv": (384, 1152, "f")}


def make(name):
    N, K, form = SHAPES[name]
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g) * 0.3
    ap, wp = ops.plane_split(rnd(M, K), 3), ops.plane_split(rnd(N, K), 3)
    bias, dact = rnd(N), rnd(M, N)
    if form == "f":
        return lambda: ops.pgemm_nt(ap, wp, bias=bias)
    if form == "p":
        return lambda: ops.pgemm_nt(ap, wp, bias=bias, out_planes=True)
    if form == "g":
        return lambda: ops.pgemm_nt(ap, wp, bias=bias, epilogue=ops.EPI_GELU, aux=True, out_planes=True)
    return lambda: ops.pgemm_nt(ap, wp, epilogue=ops.EPI_GELU_BWD, aux=dact, out_planes=True, want_colsum=True)


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    if os.environ.get("SKEW"):                  # start skew of the workgroups (msn_set_pgemm_skew, shader cycles per phase)
        from multimodal_supernovae_amd import _lib
        _lib.check(_lib.lib().msn_set_pgemm_skew(int(os.environ["SKEW"])))
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        name, cg, sr, reps = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
        fn = make(name)
        ops.set_pgemm_walk(cg, sr)
        print(f"{name} cg {cg} sr {sr}: {timed(fn, reps):.1f} us", flush=True)
        return
    names = sys.argv[1:] or list(SHAPES)
    print("lib:", os.environ.get("MSN_HIP_LIB", "default"), flush=True)
    for name in names:
        N, K, _ = SHAPES[name]
        tn = (N + 127) // 128
        fn = make(name)
        cgs = sorted({0} | {c for c in (2, 3, 4, 6) if c < tn})
        srs = [0, 1, 2, 4, 8] if K <= 384 else [0, 2]
        cfgs = [(c, s) for c in cgs for s in srs]
        for _ in range(10):
            fn()
        best = {c: [] for c in cfgs}
        for rnd_ in range(3):
            for c in cfgs:
                ops.set_pgemm_walk(*c)
                fn()
                best[c].append(timed(fn, 12))
        ops.set_pgemm_walk(0, 0)
        base = min(best[(0, 0)])
        line = f"{name:6s} N={N:4d} K={K:4d} default {base:6.1f} us |"
        for c in cfgs[1:]:
            t = min(best[c])
            line += f" cg{c[0]}/sr{c[1]} {t:6.1f} ({(t / base - 1) * 100:+.1f}%)"
        print(line, flush=True)


if __name__ == "__main__":
    main()

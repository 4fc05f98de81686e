#!/usr/bin/env python3
"""The plane NT kernel on the headline shapes in its output forms, ONE shape per process: us per launch, HIP events (the bench of
the round-6 diagnostic builds -- tools/microbench/build_ablate.sh PG_* through MSN_HIP_LIB -- and of tools/pmc_walk.sh's FETCH_SIZE /
WRITE_SIZE passes; the two numeric arguments were the tile walk's column group and super-rows while msn_set_pgemm_walk existed:
commit 03e790c, profiles/r06_experiments_tried.txt item 1 -- they are ignored now).

    python tools/bench_pgemm_walk.py --one fc1g 0 0 40     # 40 launches of fc1 with GELU and plane output"""
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
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        name, cg, sr, reps = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
        fn = make(name)
        print(f"{name} cg {cg} sr {sr}: {timed(fn, reps):.1f} us", flush=True)
        return
    for name in sys.argv[1:] or list(SHAPES):
        fn = make(name)
        for _ in range(10):
            fn()
        print(f"{name}: {min(timed(fn, 12) for _ in range(3)):.1f} us", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The reference-native towers' products (emb 32 / 64: K and N of 32 .. 256 over 200k token rows) are HBM-bound:
time of msn_sgemm per shape against the bytes it has to move (A + C [+ aux / residual]) at 8 TB/s.

    python tools/bench_gemm_skinny.py [rows_sp rows_lc]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_supernovae_amd import ops

MS, ML = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024 * 220, 1024 * 200)


def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


# (name, M, N, K, opA, opB, epilogue, extra bytes per output element read (aux / residual))
def shapes(tag, M, e, ff):
    return [
        (f"{tag} qkv fwd", M, 3 * e, e, 0, 1, "none"), (f"{tag} unify fwd (+res)", M, e, e, 0, 1, "add"),
        (f"{tag} ff1 fwd (relu)", M, ff, e, 0, 1, "relu"), (f"{tag} ff2 fwd (+res)", M, e, ff, 0, 1, "add"),
        (f"{tag} ff2 dgrad (relu')", M, ff, e, 0, 0, "relu_bwd"), (f"{tag} ff1 dgrad", M, e, ff, 0, 0, "none"),
        (f"{tag} unify dgrad", M, e, e, 0, 0, "none"), (f"{tag} qkv dgrad", M, e, 3 * e, 0, 0, "none"),
        (f"{tag} qkv wgrad", 3 * e, e, M, 1, 0, "none"), (f"{tag} ff1 wgrad", ff, e, M, 1, 0, "none"),
        (f"{tag} ff2 wgrad", e, ff, M, 1, 0, "none"), (f"{tag} unify wgrad", e, e, M, 1, 0, "none"),
    ]


if os.environ.get("MSN_GEMM_VARIANT"):
    ops.set_gemm_variant(int(os.environ["MSN_GEMM_VARIANT"]))
if os.environ.get("TILE_N"):
    ops.set_gemm_tile_n(int(os.environ["TILE_N"]))
total_t = total_f = 0.0
for name, M, N, K, oa, ob, epi in shapes("SP e32", MS, 32, 128) + shapes("LC e64", ML, 64, 256):
    a = torch.randn((M, K) if oa == 0 else (K, M), device="cuda")
    b = torch.randn((K, N) if ob == 0 else (N, K), device="cuda")
    out = torch.empty(M, N, device="cuda")
    kw = {}
    extra = 0
    if epi == "add":
        kw = dict(epilogue=ops.EPI_ADD, aux=torch.randn(M, N, device="cuda"), bias=torch.randn(N, device="cuda")); extra = M * N
    elif epi == "relu":
        kw = dict(epilogue=ops.EPI_RELU, bias=torch.randn(N, device="cuda"))
    elif epi == "relu_bwd":
        kw = dict(epilogue=ops.EPI_RELU_BWD, aux=torch.randn(M, N, device="cuda")); extra = M * N
    t = timeit(lambda: ops.sgemm(a, b, oa, ob, out=out, **kw))
    nbytes = 4 * (a.numel() + b.numel() + M * N + extra)
    floor = nbytes / 8e12 * 1e6
    total_t += t; total_f += floor
    print(f"{name:28s} M={M:7d} N={N:4d} K={K:7d}  {t:7.1f} us   {nbytes / 1e6:7.1f} MB  floor {floor:6.1f} us  frac {floor / t:4.2f}  {2 * M * N * K / t / 1e6:6.1f} TFLOP/s", flush=True)
print(f"sum {total_t:.0f} us, floor {total_f:.0f} us ({total_f / total_t:.2f})")

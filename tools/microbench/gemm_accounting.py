#!/usr/bin/env python3
"""Per-LAUNCH accounting of the fp32 GEMM (diagnostic MSN_TIMELINE build, tools/microbench/build_timeline.sh): where the
time between "all matrix pipes busy" and the measured launch goes.  For every shape the recorded launch's workgroup
timestamps (start, first K-step landed, K loop done, stores acknowledged) and hardware ids give, per CU,

  * the time with 2 / 1 / 0 workgroups resident (ramp, drain, dispatch gaps),
  * the time with at least one resident workgroup inside its K loop (the only time the matrix pipe can be fed),

and per launch the ideal matrix-pipe time of its tiles at the clock measured inside the K loop.  The table splits
span - ideal into: (a) CU time with no workgroup resident, (b) resident but nobody in the K loop (prologues / epilogues
not covered by the partner), (c) K-loop time above the ideal (pipe shared / stalled while in the loop).

usage: MSN_HIP_LIB=tools/microbench/ablate/libmsn_timeline.so python tools/microbench/gemm_accounting.py [B ...]
"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from multimodal_supernovae_amd import ops, _lib

L = _lib.lib()
L.msn_debug_timeline.restype = ctypes.c_int
L.msn_debug_timeline.argtypes = [ctypes.c_void_p]
NBLK = 1 << 16
dbg = torch.zeros(NBLK * 10, dtype=torch.int64, device="cuda")


def union_length(iv):
    """total length of the union of intervals [(a, b), ...]"""
    if not iv:
        return 0.0
    iv = sorted(iv)
    tot, (ca, cb) = 0.0, iv[0]
    for a, b in iv[1:]:
        if a > cb:
            tot += cb - ca
            ca, cb = a, b
        else:
            cb = max(cb, b)
    return tot + (cb - ca)


def occupancy(iv, span):
    """time with exactly 0 / 1 / >= 2 of the intervals open, over [0, span]"""
    ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
    t_prev, n, occ = 0.0, 0, [0.0, 0.0, 0.0]
    for t, d in ev:
        occ[min(n, 2)] += t - t_prev
        t_prev, n = t, n + d
    occ[0] += span - t_prev
    return occ


def record(M, N, K, oa, ob, warm=30):
    a = torch.randn((M, K) if oa == 0 else (K, M), device="cuda")
    b = torch.randn((K, N) if ob == 0 else (N, K), device="cuda")
    for _ in range(warm):
        ops.sgemm(a, b, oa, ob)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.sgemm(a, b, oa, ob)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    dbg.zero_()
    L.msn_debug_timeline(dbg.data_ptr())
    ops.sgemm(a, b, oa, ob)
    torch.cuda.synchronize()
    L.msn_debug_timeline(None)
    d = dbg.cpu().numpy().reshape(-1, 10)
    d = d[d[:, 0] != 0]
    return us, d


def account(tag, M, N, K, oa, ob):
    us, d = record(M, N, K, oa, ob)
    t = (d[:, :4] - d[:, 0].min()) * 0.01           # us (100 MHz real-time counter)
    span = t[:, 3].max()
    hw, xcc = d[:, 4], d[:, 5] & 0xF
    cu = [(int(x), int((h >> 13) & 7), int((h >> 8) & 0xF)) for x, h in zip(xcc, hw)]
    ghz = float((d[:, 9] / np.maximum((d[:, 2] - d[:, 1]) * 10.0, 1.0)).mean())
    per = {}
    for i, c in enumerate(cu):
        per.setdefault(c, []).append(i)
    ncu = 256
    occ = np.zeros(3)
    feed = 0.0
    for c, idx in per.items():
        occ += occupancy([(t[i, 0], t[i, 3]) for i in idx], span)
        feed += union_length([(t[i, 1], t[i, 2]) for i in idx])
    occ[0] += (ncu - len(per)) * span               # CUs that never saw a workgroup
    # ideal matrix-pipe time: 2 M N K flop at the guide's fp32 matrix peak (157.3 TFLOP/s at 2.4 GHz) scaled to the clock
    # measured inside the K loop
    peak_tflops = 157.3 * ghz / 2.4
    ideal = 2.0 * M * N * K / peak_tflops / 1e6     # us
    cu_span = ncu * span
    resident_nofeed = cu_span - occ[0] - feed
    loop_excess = feed - ideal * ncu
    wg_pro, wg_loop, wg_epi = (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 3] - t[:, 2]).mean()
    print(f"{tag:16s} M={M:6d} N={N:5d} K={K:6d} op={oa}{ob} | {us:7.1f} us ({2.0 * M * N * K / us / 1e6:6.1f} TF) span {span:7.1f} "
          f"wgs {len(d):5d} clk {ghz:.2f} | ideal {ideal:6.1f} ({100 * ideal / span:4.1f}%) | CU time: empty {100 * occ[0] / cu_span:4.1f}% "
          f"one wg {100 * occ[1] / cu_span:4.1f}% two {100 * occ[2] / cu_span:4.1f}% | nobody in K loop {100 * resident_nofeed / cu_span:4.1f}% "
          f"K-loop above ideal {100 * loop_excess / cu_span:4.1f}% | per wg: pro {wg_pro:4.1f} loop {wg_loop:5.1f} epi {wg_epi:4.1f} us", flush=True)


if __name__ == "__main__":
    batches = [int(v) for v in sys.argv[1:]] or [128, 1024]
    for B in batches:
        M = B * 65
        print(f"--- headline ViT-S/8 products at {B} rows per GPU (M = {M})")
        for N, K, oa, ob, tag in [(384, 384, 0, 1, "proj fwd"), (1152, 384, 0, 1, "qkv fwd"), (1536, 384, 0, 1, "ff1 fwd"),
                                  (384, 1536, 0, 1, "ff2 fwd"), (384, 1152, 0, 0, "qkv dgrad"), (384, 1536, 0, 0, "ff1 dgrad"),
                                  (1536, 384, 0, 0, "ff2 dgrad"), (384, 384, 0, 0, "proj dgrad")]:
            account(tag, M, N, K, oa, ob)
        for Mw, Nw, tag in [(384, 384, "proj wgrad"), (1152, 384, "qkv wgrad"), (1536, 384, "ff1 wgrad"), (384, 1536, "ff2 wgrad")]:
            account(tag, Mw, Nw, M, 1, 0)

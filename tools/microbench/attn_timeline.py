#!/usr/bin/env python3
"""Where a workgroup of the chunked matrix-core attention forward spends its time (diagnostic build with -DMSN_ATTN_TIMELINE,
tools/microbench/ablate/libmsn_attn_timeline.so): shader-clock stamps of wave 0 of every workgroup -- entry, requests issued,
rows in LDS, barrier passed, multiplication done, stores drained -- averaged over the workgroups of one launch, and how many
workgroups a CU held at once (from the stamps and HW_ID).
    MSN_HIP_LIB=tools/microbench/ablate/libmsn_attn_timeline.so python tools/microbench/attn_timeline.py"""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from multimodal_supernovae_amd import ops, _lib

L = _lib.lib()
for name, B, T, E, H in [("spectrum tower", 1024, 220, 32, 2), ("light-curve tower (8-wide heads as 16)", 1024, 200, 64, 8)]:
    qkv = torch.randn(B, T, 3 * E, device="cuda")
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    mask = torch.arange(T, device="cuda")[None, :] < torch.randint(T // 2, T + 1, (B,), device="cuda")[:, None]
    L.msn_set_attention_path(2)
    wgs = B * H * ((T + 127) // 128)
    buf = np.zeros((wgs, 8), dtype=np.uint64)
    for _ in range(3):
        ops.attention_fwd(q, k, v, mask, H, 1 / math.sqrt(E))
    torch.cuda.synchronize()
    assert L.msn_mattn_debug_read(buf.ctypes.data_as(ctypes.c_void_p), wgs) == 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.attention_fwd(q, k, v, mask, H, 1 / math.sqrt(E))
    e.record(); torch.cuda.synchronize()
    assert L.msn_mattn_debug_read(buf.ctypes.data_as(ctypes.c_void_p), wgs) == 0
    t = buf[:, :6].astype(np.int64)
    d = np.diff(t, axis=1).mean(axis=0)
    span = (t[:, 5].max() - t[:, 0].min())
    life = (t[:, 5] - t[:, 0]).sum()
    print(f"{name}: {s.elapsed_time(e) * 1e3:.1f} us, {wgs} workgroups; cycles (wave 0, mean): requests issued {d[0]:.0f} | rows arrived + in LDS {d[1]:.0f} | "
          f"barrier {d[2]:.0f} | multiply {d[3]:.0f} | normalise + store + drain {d[4]:.0f} | lifetime {(t[:, 5] - t[:, 0]).mean():.0f}; "
          f"launch span {span} cycles -> {life / span / 256:.2f} workgroups resident per CU on average")
L.msn_set_attention_path(0)

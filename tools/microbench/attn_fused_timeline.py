#!/usr/bin/env python3
"""Stamps of wave 0 of every workgroup of mattn_bwd_fused_kernel (diagnostic build, see attn_fused_timeline.sh): mean cycles per
phase and the number of workgroups a CU held at once."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from multimodal_supernovae_amd import ops, _lib

L = _lib.lib()
B, T, H, hd = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (1024, 65, 6, 64)
E = H * hd
qkv = torch.randn(B, T, 3 * E, device="cuda")
dout = torch.randn(B, T, E, device="cuda")
scale = 1 / math.sqrt(hd)
out, lse = ops.attention_fwd(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], None, H, scale)
wgs = B * H
buf = np.zeros((wgs, 8), dtype=np.uint64)
for planes in (True, False):
    def run():
        if planes:
            ops.attention_bwd_planes(qkv, H, scale, out, lse, dout, 3)
        else:
            d = torch.empty_like(qkv)
            ops.attention_bwd(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], None, H, scale, out, lse, dout, d[..., :E], d[..., E:2 * E], d[..., 2 * E:])
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    assert L.msn_mattn_debug_read(buf.ctypes.data_as(ctypes.c_void_p), wgs) == 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    run()
    e.record(); torch.cuda.synchronize()
    assert L.msn_mattn_debug_read(buf.ctypes.data_as(ctypes.c_void_p), wgs) == 0
    t = buf.astype(np.int64)
    d = np.diff(t, axis=1).mean(axis=0)
    span = t[:, 7].max() - t[:, 0].min()
    life = (t[:, 7] - t[:, 0]).sum()
    print(f"{'planes' if planes else 'fp32 rows'}: B={B} T={T} H={H} hd={hd}: {s.elapsed_time(e) * 1e3:.1f} us, {wgs} workgroups; cycles (wave 0, mean): "
          f"requests issued {d[0]:.0f} | rows arrived, committed, barrier {d[1]:.0f} | delta + barrier {d[2]:.0f} | dQ phase {d[3]:.0f} | dK,dV phase {d[4]:.0f} | "
          f"wait for the other waves {d[5]:.0f} | stage + store {d[6]:.0f} | lifetime {(t[:, 7] - t[:, 0]).mean():.0f}; launch span {span} cycles -> "
          f"{life / span / 256:.2f} workgroups resident per CU on average")

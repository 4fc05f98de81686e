#!/usr/bin/env python3
"""Per-workgroup timeline of one msn_sgemm launch (diagnostic build: `bash tools/microbench/build_timeline.sh [1|2]`
compiles csrc/gemm.hip with -DMSN_TIMELINE into tools/microbench/ablate/libmsn_timeline.so).  Prints where a tile's time goes: start -> first K-step landed ->
K loop done -> stores acknowledged, and how the workgroups spread over XCDs / CUs / rounds.
usage: MSN_HIP_LIB=.../libmsn_timeline.so python tools/microbench/gemm_timeline.py M N K opA opB [warm-up launches]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from multimodal_supernovae_amd import ops, _lib

M, N, K, oa, ob = [int(v) for v in sys.argv[1:6]]
a = torch.randn((M, K) if oa == 0 else (K, M), device="cuda")
b = torch.randn((K, N) if ob == 0 else (N, K), device="cuda")
L = _lib.lib()
L.msn_debug_timeline.restype = ctypes.c_int
L.msn_debug_timeline.argtypes = [ctypes.c_void_p]
nblk = 1 << 16
dbg = torch.zeros(nblk * 10, dtype=torch.int64, device="cuda")
warm = int(sys.argv[6]) if len(sys.argv) > 6 else 3      # launches before the recorded one (sustained-load clock)
for _ in range(warm):
    ops.sgemm(a, b, oa, ob)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.sgemm(a, b, oa, ob)
e1.record()
torch.cuda.synchronize()
print(f"{2.0 * M * N * K * 20 / e0.elapsed_time(e1) / 1e9:.1f} TFLOP/s over 20 launches after {warm} warm-up launches")
L.msn_debug_timeline(dbg.data_ptr())
ops.sgemm(a, b, oa, ob)
torch.cuda.synchronize()
L.msn_debug_timeline(None)
d = dbg.cpu().numpy().reshape(-1, 10)
d = d[d[:, 0] != 0]
t = (d[:, :4] - d[:, 0].min()) * 0.01          # us (100 MHz)
hw, xcc = d[:, 4], d[:, 5] & 0xF
cu, se = (hw >> 8) & 0xF, (hw >> 13) & 0x7
print(f"{len(d)} workgroups; kernel span {t[:, 3].max():.1f} us")
loop = d[:, 9].astype(float)
print(f"core clock inside the K loop: {(loop / ((d[:,2] - d[:,1]) * 10.0)).mean():.3f} GHz (s_memtime / s_memrealtime)")
print(f"wave 0 inside the K loop: {loop.mean():.0f} cycles; waiting for fragments {100 * (d[:,6] / loop).mean():.1f} %, for its LDS-DMA pieces {100 * (d[:,7] / loop).mean():.1f} %, at the barrier {100 * (d[:,8] / loop).mean():.1f} %")
print(f"mean per workgroup: prologue {np.mean(t[:,1]-t[:,0]):.2f} us | K loop {np.mean(t[:,2]-t[:,1]):.2f} us | epilogue {np.mean(t[:,3]-t[:,2]):.2f} us")
order = np.argsort(t[:, 0])
starts = t[order, 0]
print("start-time deciles (us):", np.round(np.percentile(starts, np.arange(0, 101, 10)), 1))
print("end-time deciles   (us):", np.round(np.percentile(t[:, 3], np.arange(0, 101, 10)), 1))
first = d[:512]
slot = {}
for i in range(min(len(d), 512)):
    slot.setdefault((int(xcc[i]), int(se[i]), int(cu[i])), []).append(i)
print("distinct (xcc, se, cu) among the first 512 workgroups:", len(slot), "| max per CU:", max(len(v) for v in slot.values()))
for lo in range(0, min(len(d), 2048), 512):
    sel = slice(lo, lo + 512)
    print(f"wg {lo:5d}..: start {t[sel,0].mean():7.1f}  prologue {np.mean(t[sel,1]-t[sel,0]):5.2f}  loop {np.mean(t[sel,2]-t[sel,1]):6.2f}  epilogue {np.mean(t[sel,3]-t[sel,2]):5.2f}")
key = [(int(xcc[i]), int(se[i]), int(cu[i])) for i in range(len(d))]
#print("wg -> (xcc, se, cu):", [(i, key[i]) for i in list(range(0, 20)) + list(range(254, 262)) + list(range(510, 516))])
pairs = {}
for i in range(min(len(d), 512)):
    pairs.setdefault(key[i], []).append(i)
diffs = [v[1] - v[0] for v in pairs.values() if len(v) == 2]
print("index distance between the two first-round workgroups of a CU: min", min(diffs), "max", max(diffs), "histogram", np.unique(diffs, return_counts=True))
# who follows whom: for each CU, the sequence of workgroup ids it ran
seq = {}
for i in np.argsort(t[:, 0]):
    seq.setdefault(key[i], []).append(int(i))
k0 = sorted(seq)[0]
print("CU", k0, "ran:", seq[k0][:12])
print("hw_id bits of wg0:", hex(int(hw[0])), "wave slot", int(hw[0]) & 0xF, "simd", (int(hw[0]) >> 4) & 3)

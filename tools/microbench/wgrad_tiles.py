import os, sys
sys.path.insert(0, os.getcwd())
import torch
from multimodal_supernovae_amd import ops
def timed(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for K in (8320, 16640, 33280, 66560):
    for (M, N) in ((384, 384), (1152, 384), (384, 1536), (1536, 384)):
        dy = torch.randn(K, M, device="cuda"); x = torch.randn(K, N, device="cuda")
        res = []
        for bn in (0, 64):
            ops.set_gemm_tile_n(bn)
            t = timed(lambda: ops.sgemm(dy, x, op_a=ops.OP_T, op_b=ops.OP_N))
            res.append(t)
        ops.set_gemm_tile_n(0)
        print(f"wgrad {M}x{N} K={K}: auto {res[0]*1e6:7.1f} us ({2.0*M*N*K/res[0]/1e12:6.1f} TF)   bn=64 {res[1]*1e6:7.1f} us ({2.0*M*N*K/res[1]/1e12:6.1f} TF)", flush=True)

#!/usr/bin/env python3
"""The fused feed-forward kernels of the narrow towers (msn_ffn_fwd / _bwd, emb 32) against the four-product path they replace
(msn_sgemm + epilogues, msn_wgrad_bias), us per layer, at the token counts of the reference-native towers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_supernovae_amd import ops

def timeit(fn, iters=30, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

E_, HID = 32, 128
for M, what in [(1024 * 220, "Maven spectrum tower (1024 x 220 tokens)"), (1024 * 1024, "1024-bin spectra (1024 x 1024 tokens)"), (256 * 220, "256 x 220 tokens")]:
    g = torch.Generator(device="cuda").manual_seed(0)
    x, dz = torch.randn(M, E_, device="cuda", generator=g), torch.randn(M, E_, device="cuda", generator=g)
    w1, w2 = torch.randn(HID, E_, device="cuda", generator=g) * 0.2, torch.randn(E_, HID, device="cuda", generator=g) * 0.1
    c1, c2 = torch.randn(HID, device="cuda", generator=g) * 0.1, torch.randn(E_, device="cuda", generator=g) * 0.1
    P = ops.PREC_F32
    def unf_fwd():
        hdn = ops.sgemm(x, w1, ops.OP_N, ops.OP_T, bias=c1, epilogue=ops.EPI_RELU, precision=P)
        return hdn, ops.sgemm(hdn, w2, ops.OP_N, ops.OP_T, bias=c2, epilogue=ops.EPI_ADD, aux=x, precision=P)
    hdn, _ = unf_fwd()
    def unf_bwd():
        ops.wgrad_bias(dz, hdn, precision=P)
        dpre = ops.sgemm(dz, w2, ops.OP_N, ops.OP_N, epilogue=ops.EPI_RELU_BWD, aux=hdn, precision=P)
        ops.wgrad_bias(dpre, x, precision=P)
        ops.sgemm(dpre, w1, ops.OP_N, ops.OP_N, epilogue=ops.EPI_ADD, aux=dz, precision=P)
    w1p, w2tp = ops.ffn_weight_planes(w1, w2)
    t = [timeit(unf_fwd), timeit(unf_bwd), timeit(lambda: ops.ffn_weight_planes(w1, w2)), timeit(lambda: ops.ffn_fwd(x, w1p, w2tp, c1, c2)),
         timeit(lambda: ops.ffn_bwd(x, dz, w1p, w2tp, c1))]
    print(f"{what}: unfused fwd {t[0]:.0f} + bwd {t[1]:.0f} = {t[0] + t[1]:.0f} us | fused: weight planes {t[2]:.0f} + fwd {t[3]:.0f} + bwd {t[4]:.0f} = {sum(t[2:]):.0f} us", flush=True)

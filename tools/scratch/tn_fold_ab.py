#!/usr/bin/env python3
"""A/B of the weight-gradient kernel's fold (MSN_TN_FOLD2, experimental library only): time on the four headline shapes, error
against fp64 next to the native fp32 kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multimodal_supernovae_amd import ops


def timed(fn, reps=40):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


M = 1024 * 65
g = torch.Generator(device="cuda").manual_seed(0)
for kind in ("normal", "positive"):
    for name, N, K in [("wqkv", 1152, 384), ("wo", 384, 384), ("w1", 1536, 384), ("w2", 384, 1536)]:
        dy = torch.randn(M, N, device="cuda", generator=g) * 0.3
        x = torch.randn(M, K, device="cuda", generator=g) * 0.3
        if kind == "positive":
            dy, x = dy.abs(), x.abs()
        ref = dy.double().t() @ x.double()
        nat = ops.wgrad_bias(dy, x, precision=ops.PREC_F32)
        nat = nat[0] if isinstance(nat, (tuple, list)) else nat
        dp, xp = ops.plane_split(dy, 3), ops.plane_split(x, 3)
        line = f"{kind:8s} {name:5s}"
        en = (nat.double() - ref)
        line += f" native max {en.abs().max().item():.3e} rms {en.pow(2).mean().sqrt().item():.3e}"
        for fold2 in (0, 1, 0, 1):
            if fold2:
                os.environ["MSN_TN_FOLD2"] = "1"
            else:
                os.environ.pop("MSN_TN_FOLD2", None)
            out = ops.pgemm_tn(dp, xp)
            e = out.double() - ref
            t = timed(lambda: ops.pgemm_tn(dp, xp))
            line += f" | fold{fold2 + 1} {t:6.0f} us max {e.abs().max().item() / en.abs().max().item():.2f}x rms {e.pow(2).mean().sqrt().item() / en.pow(2).mean().sqrt().item():.2f}x"
        print(line, flush=True)

import os
import sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multimodal_supernovae_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
M, N = 2080, 384
for variant in (0, 1):
    ops.set_pgemm_variant(variant)
    for K in (384, 1536, 6144):
        a, w = torch.randn(M, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g)
        ref = a.double() @ w.double().T
        wp = ops.plane_split(w, 3)
        c1 = ops.pgemm_nt(ops.plane_split(a, 3), wp).double()
        c2 = -ops.pgemm_nt(ops.plane_split(-a, 3), wp).double()
        e1, e2, em = c1 - ref, c2 - ref, 0.5 * (c1 + c2) - ref
        pos = ref > 0
        print(f"v{variant} K={K}: mean err {e1.mean():+.3e}  mean err of -(−A·B) {e2.mean():+.3e}   rms {e1.pow(2).mean().sqrt():.3e} / {e2.pow(2).mean().sqrt():.3e}"
              f"  rms of the average {em.pow(2).mean().sqrt():.3e};  mean err where ref>0 {e1[pos].mean():+.3e}, ref<0 {e1[~pos].mean():+.3e}")

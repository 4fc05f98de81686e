import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_supernovae_amd import ops, _lib
torch.manual_seed(0)
B, T, H, hd = 1, 160, 1, 16
E = H * hd
q, k, v = (torch.randn(B, T, E, device="cuda") for _ in range(3))
scale = 1 / math.sqrt(E)
S = (q[0].double() @ k[0].double().T) * scale          # [query][key]
for mode in (1, 3):
    ops.set_attention_planes(mode)
    out, lse = ops.attention_fwd(q, k, v, None, H, scale)
    m = lse[0, 0, :, 0].double()
    true_m = S.max(1).values
    bad = ((m - true_m).abs() > 1e-4).nonzero().flatten().tolist()
    print("mode", mode, "bad rows", len(bad), bad[:40])
    for r in bad[:6]:
        # which key's score equals my m?
        d = (S[r] - m[r]).abs()
        j = int(d.argmin())
        print("  row", r, "my m", float(m[r]), "true", float(true_m[r]), "argmax", int(S[r].argmax()), "closest key", j, "diff", float(d[j]))
    P = torch.softmax(S, 1)
    ref = P @ v[0].double()
    print("  out max err", float((out[0].double() - ref).abs().max()))
    # m as max over a subset?  for each bad row list keys whose score > my m
    for r in bad[:6]:
        print("  row", r, "keys above my m:", (S[r] > m[r] + 1e-4).nonzero().flatten().tolist()[:20])

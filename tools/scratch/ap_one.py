import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_supernovae_amd import ops, _lib
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B, T, E, H = 1024, 1024, 32, 2
qkv = torch.randn(B, T, 3 * E, device="cuda")
q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
dout = torch.randn(B, T, E, device="cuda"); dqkv = torch.empty_like(qkv)
scale = 1 / math.sqrt(E)
ops.set_attention_planes(mode)
for _ in range(3):
    out, lse = ops.attention_fwd(q, k, v, None, H, scale)
    ops.attention_bwd(q, k, v, None, H, scale, out, lse, dout, dqkv[..., :E], dqkv[..., E:2 * E], dqkv[..., 2 * E:])
torch.cuda.synchronize()

"""Self-attention backward at the ViT towers' shapes: the dQ + dK,dV kernel pair, the one-pass kernel (fp32 rows), and the
one-pass kernel writing dqkv as planes (with / without the column sums), next to msn_plane_split of the fp32 gradient.
    python tools/bench_attention_fused.py [B T heads hd]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_supernovae_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B, T, heads, hd = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (1024, 65, 6, 64)
    E = heads * hd
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B, T, 3 * E, generator=g).cuda()
    dout = torch.randn(B, T, E, generator=g).cuda()
    scale = 1.0 / math.sqrt(hd)
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    out, lse = ops.attention_fwd(q, k, v, None, heads, scale)
    dqkv = torch.empty_like(qkv)

    def pair():
        ops.attention_bwd(q, k, v, None, heads, scale, out, lse, dout, dqkv[..., :E], dqkv[..., E:2 * E], dqkv[..., 2 * E:])

    ops.set_attention_fused(False)
    t_pair = timeit(pair)
    ops.set_attention_fused(True)
    t_one = timeit(pair)
    t_split = timeit(lambda: ops.plane_split(dqkv.view(B * T, 3 * E), 3, want_colsum=True))
    t_pl = timeit(lambda: ops.attention_bwd_planes(qkv, heads, scale, out, lse, dout, 3, want_colsum=False))
    t_plc = timeit(lambda: ops.attention_bwd_planes(qkv, heads, scale, out, lse, dout, 3, want_colsum=True))
    t_fwd = timeit(lambda: ops.attention_fwd(q, k, v, None, heads, scale))
    print(f"B={B} T={T} heads={heads} hd={hd}: fwd {t_fwd:.0f} us | bwd: two kernels {t_pair:.0f} us, one pass {t_one:.0f} us; "
          f"split+colsum {t_split:.0f} us; one pass -> planes {t_pl:.0f} us, + column sums {t_plc:.0f} us")


if __name__ == "__main__":
    main()

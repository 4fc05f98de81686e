"""LayerNorm forward / backward writing planes at the headline shape: the row-block kernel (whole plane images, contiguous stores)
against the row-at-a-time kernel, and the fp32-output kernel for scale.    python tools/bench_ln_planes.py [rows cols]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_supernovae_amd import _lib, ops  # noqa: E402


def timeit(fn, n=30):
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


rows, cols = (int(x) for x in sys.argv[1:3]) if len(sys.argv) >= 3 else (66560, 384)
x = torch.randn(rows, cols, device="cuda")
gm, bt = torch.ones(cols, device="cuda"), torch.zeros(cols, device="cuda")
L = _lib.lib()
t32 = timeit(lambda: ops.layernorm_fwd(x, gm, bt, 1e-6))
tp = timeit(lambda: ops.layernorm_fwd_planes(x, gm, bt, 1e-6, 3))
mb = rows * cols * (4 + 6) / 1e6
print(f"rows={rows} cols={cols}: fp32 out {t32:.1f} us | planes ({'row blocks' if rows >= 32768 and 260 <= cols <= 400 else 'row at a time'}): {tp:.1f} us ({mb / tp:.2f} TB/s)")

# backward: dx as fp32 AND planes, residual gradient added, column sums (the call of the ViT blocks' backward)
dy, add = torch.randn(rows, cols, device="cuda"), torch.randn(rows, cols, device="cuda")
_, mean, rstd = ops.layernorm_fwd(x, gm, bt, 1e-6)
tb = timeit(lambda: ops.layernorm_bwd_planes(dy, x, mean, rstd, gm, 3, add=add, want_colsum=True))
mbb = rows * cols * (4 * 3 + 4 + 6) / 1e6
print(f"backward (x, dy, add -> dx fp32 + 3 planes): {tb:.1f} us ({mbb / tb:.2f} TB/s)")

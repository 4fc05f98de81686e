#!/usr/bin/env python3
"""Fused InfoNCE forward + backward latency (HIP events around the C-ABI calls, warm): the loss is latency-bound
(0.8 GFLOP at N = 1024, D = 128), so what is reported is microseconds per call, not TFLOP/s."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_supernovae_amd.loss import HipPairKernels as K  # noqa: E402


def unit(n, d, seed):
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(seed))
    return (x / x.norm(dim=-1, keepdim=True)).cuda()


def main():
    for b, n, d in [(1024, 1024, 128), (256, 256, 128), (32, 32, 128), (128, 1024, 128), (256, 2048, 128), (4096, 4096, 128),
                    (1024, 1024, 100)]:
        e1, e2 = unit(n, d, 1), unit(n, d, 2)
        l1, l2 = e1[:b], e2[:b]
        ls, lb, one = torch.tensor(math.log(19.5)).cuda(), torch.tensor(-10.0).cuda(), torch.tensor(1.0).cuda()
        lr, lc, _ = K.forward(e1, e2, e1, e2, 0, ls, lb)       # full LSE vectors for the sharded backward
        for _ in range(20):
            K.forward(l1, l2, e1, e2, 0, ls, lb)
            K.backward(l1, l2, e1, e2, 0, ls, lb, lr, lc, one)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        reps = 200
        ev[0].record()
        for _ in range(reps):
            K.forward(l1, l2, e1, e2, 0, ls, lb)
        ev[1].record()
        for _ in range(reps):
            K.backward(l1, l2, e1, e2, 0, ls, lb, lr, lc, one)
        ev[2].record()
        torch.cuda.synchronize()
        f, bw = ev[0].elapsed_time(ev[1]) / reps * 1e3, ev[1].elapsed_time(ev[2]) / reps * 1e3
        print(f"b={b:5d} N={n:5d} D={d:4d}: fwd {f:7.1f} us  bwd {bw:7.1f} us  fwd+bwd {f + bw:7.1f} us "
              f"(incl. host launch gaps and the workspace allocations of the Python wrapper)", flush=True)


if __name__ == "__main__":
    main()

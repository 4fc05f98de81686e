"""RAdam with torch.optim.RAdam's semantics and state layout, stepped by ONE fused HIP launch.

The reference builds `torch.optim.RAdam(self.parameters(), lr=..., **optimizer_kwargs)` in
configure_optimizers (src/models_multimodal.py:306-310) with torch defaults; this class keeps the
constructor, `param_groups`, `state` keys (`step`, `exp_avg`, `exp_avg_sq`), `zero_grad` and
`step`, so optimiser states of reference checkpoints map one to one.
"""
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


class RAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid RAdam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _staging(self, n):
        """Two pinned buffers used alternately; a buffer is rewritten only after the copy that last read it
        has completed (its event), so the host may run a whole step ahead of the GPU."""
        slots = getattr(self, "_pinned", None)
        if slots is None:
            slots = self._pinned = [[None, None], [None, None]]
            self._slot = 0
        self._slot ^= 1
        slot = slots[self._slot]
        if slot[1] is not None:
            slot[1].synchronize()
        if slot[0] is None or slot[0].numel() < n:
            slot[0] = torch.empty(max(n, 1024), dtype=torch.int64).pin_memory()
        return slot

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            # parameters of one group that share a step count go into one launch
            buckets = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda":
                    _lib.require_gpu()
                    raise _lib.MsnHipError("RAdam parameters must live on the GPU")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.MsnHipError("RAdam supports contiguous float32 parameters only")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1
                buckets.setdefault((st["step"], p.device), []).append((p, st))
            for (step, dev), items in buckets.items():
                words, max_n = [], 0
                keep = []
                for p, st in items:
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    keep.append(g)
                    words += [p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()]
                    max_n = max(max_n, p.numel())
                # descriptor table: persistent pinned staging buffer + async copy, so the step never
                # blocks the host on the stream (a pageable H2D copy would drain the whole queue)
                slot = self._staging(len(words))
                slot[0][:len(words)] = torch.tensor(words, dtype=torch.int64)
                table = slot[0][:len(words)].to(dev, non_blocking=True)
                slot[1] = torch.cuda.Event()
                slot[1].record()
                b1, b2 = group["betas"]
                check(lib().msn_radam_step(ptr(table), len(items), max_n, group["lr"], b1, b2, group["eps"],
                                           group["weight_decay"], step, stream_ptr()), "msn_radam_step")
        return loss

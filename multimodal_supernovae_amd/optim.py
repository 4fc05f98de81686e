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

    def _init_state(self):
        """Moment buffers of every parameter that has a gradient and no state yet, as views of ONE zeroed buffer per
        device (one fill launch instead of two per parameter; the step's launch reads them through the pointer table)."""
        fresh = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None and len(self.state[p]) == 0 and p.device.type == "cuda" and p.dtype == torch.float32:
                    fresh.setdefault(p.device, []).append(p)
        for dev, ps in fresh.items():
            n = sum((p.numel() + 3) // 4 * 4 for p in ps)            # 16-byte aligned slices
            flat = torch.zeros(2 * n, dtype=torch.float32, device=dev)
            off = 0
            for p in ps:
                m = p.numel()
                st = self.state[p]
                st["step"] = 0
                st["exp_avg"] = flat[off:off + m].view(p.shape)
                st["exp_avg_sq"] = flat[n + off:n + off + m].view(p.shape)
                off += (m + 3) // 4 * 4

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

    # ---- HIP-graph capture (trainer.GraphedTrainStep) --------------------------------------------------------------
    # Under stream capture the launch is recorded through msn_radam_step_dev: the descriptor table is copied by a copy
    # node of the graph from a pinned buffer that never changes, the hyper-parameters sit in device memory, and the
    # step-dependent terms are derived on the device from a device-resident step counter that every replay increments
    # -- no host write between replays (it would race with a replay still in flight).
    def _step_captured(self):
        self._graph_launches = []
        for group in self.param_groups:
            items = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    raise _lib.MsnHipError("capture the training step after at least one eager optimizer step "
                                           "(the moment buffers must exist)")
                items.append((p, st))
            if not items:
                continue
            steps = {int(st["step"]) for _, st in items}
            if len(steps) != 1:
                raise _lib.MsnHipError("graph capture needs one step count per parameter group")
            dev = items[0][0].device
            words, max_n = [], 0
            for p, st in items:
                if not p.grad.is_contiguous():
                    raise _lib.MsnHipError("graph capture needs contiguous gradients")
                words += [p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()]
                max_n = max(max_n, p.numel())
            hyper, counter, table_host = self._graph_buffers()
            table_host = table_host[:len(words)]                      # pinned before the capture began
            table_host.copy_(torch.tensor(words, dtype=torch.int64))
            table = table_host.to(dev, non_blocking=True)             # a copy node of the graph (static content)
            check(lib().msn_radam_step_dev(ptr(table), len(items), max_n, ptr(hyper), ptr(counter), stream_ptr()),
                  "msn_radam_step_dev")
            self._graph_launches.append((group, items, table_host, table, hyper, counter))

    def graph_prepare(self):
        """Call BEFORE the capture (eager): device copies of the hyper-parameters and of the step count per group."""
        self._graph_ready = {}
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if len(self.state[p])]
            if not ps:
                continue
            b1, b2 = group["betas"]
            dev = ps[0].device
            hyper = torch.tensor([group["lr"], b1, b2, group["eps"], group["weight_decay"], 0.0, 0.0, 0.0],
                                 dtype=torch.float32, device=dev)
            counter = torch.tensor([int(self.state[ps[0]]["step"])], dtype=torch.int64, device=dev)
            table_host = torch.empty(5 * len(group["params"]), dtype=torch.int64).pin_memory()
            self._graph_ready[gi] = (hyper, counter, table_host)
            self._graph_hyper_captured = getattr(self, "_graph_hyper_captured", {})
            self._graph_hyper_captured[id(group)] = self._hyper_of(group)
        torch.cuda.synchronize()

    def _graph_buffers(self):
        ready = getattr(self, "_graph_ready", {})
        if not ready:
            raise _lib.MsnHipError("RAdam.graph_prepare() must run before the training step is captured")
        return ready.pop(min(ready))

    def graph_note_eager_step(self):
        """An eager step() ran between two replays (a batch of another shape): advance the device counters with it."""
        for _, _, _, _, _, counter in getattr(self, "_graph_launches", []):
            counter.add_(1)

    @staticmethod
    def _hyper_of(group):
        b1, b2 = group["betas"]
        return (float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]))

    def graph_pre_replay(self):
        """Keep the host-side step counts in line with the device counter a replay increments, and carry a changed
        learning rate / betas / eps / weight decay (an lr scheduler, a manual edit of param_groups) into the device
        copy the recorded launch reads: the copy is enqueued on the replaying stream BEFORE the replay, so it is
        ordered against the previous replay's read and this replay's."""
        seen = getattr(self, "_graph_hyper_seen", None)
        if seen is None:
            seen = self._graph_hyper_seen = {}
        for li, (group, items, _, _, hyper, _) in enumerate(self._graph_launches):
            step = int(items[0][1]["step"]) + 1
            for _, st in items:
                st["step"] = step
            now = self._hyper_of(group)
            if seen.setdefault(li, self._graph_hyper_captured.get(id(group), now)) != now:
                hyper[:5].copy_(torch.tensor(now, dtype=torch.float32), non_blocking=False)
                seen[li] = now

    @torch.no_grad()
    def step(self, closure=None):
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            self._step_captured()
            return None
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._init_state()
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

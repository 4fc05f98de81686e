"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" is
RCCL on ROCm) -- or gloo on CPU for the algebra tests.  The reference has no distributed code at
all; what is exchanged and why:

  * embeddings: all-gather of the local (b, D) rows per modality, then of the per-row LSEs
    (loss.py) -> every rank evaluates its own rows / columns of the global logit matrix;
  * parameter gradients: each rank's backward yields d(global loss)/d(params) through ITS samples,
    so the true gradient is the SUM over ranks (not the mean): bucketed all-reduce(SUM).
    `GradientReducer` launches a bucket's all-reduce from autograd hooks the moment its last gradient
    exists, so RCCL's transfers over xGMI run under the rest of backward; `allreduce_gradients` is the
    plain after-backward form of the same reduction.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise from torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns
    (rank, local_rank, world_size).  A no-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        # more ranks than GPUs (a 1-GPU test box running the 2-rank flow over gloo) share devices round-robin
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("MSN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local, world


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


@torch.no_grad()
def broadcast_module(module, src=0, group=None):
    """Same initial weights and buffers everywhere (rank `src` wins)."""
    if world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def enable_sync_batchnorm(group=None, enabled=True):
    """Batch statistics of every BatchNorm on the path (ConvMixer / ResNet-18 / 1-D CNN towers) over the rows of ALL
    ranks of `group` instead of per replica -- the counterpart of Lightning's `Trainer(sync_batchnorm=True)`; with it a
    data-parallel step equals the single-process step at the global batch.  The kernels are
    the single-process two-pass statistics cut where C (forward, twice) or 2C (backward) floats are all-reduced
    (ops.batchnorm_fwd / batchnorm_bwd).  Equal rows per rank are assumed.  Off by default."""
    from . import ops
    ops.BN_SYNC_GROUP = (group if group is not None else True) if enabled else None


@torch.no_grad()
def allreduce_gradients(params, group=None, bucket_bytes=32 << 20):
    """SUM-all-reduce the .grad of `params` in flat buckets of ~bucket_bytes.  Every rank must call it
    with the same parameter order; parameters without a gradient contribute zeros."""
    if world_size(group) == 1:
        return
    params = [p for p in params if p.requires_grad]
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        cur.append(p)
        cur_bytes += p.numel() * p.element_size()
        if cur_bytes >= bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
    if cur:
        buckets.append(cur)
    pending = []
    for bucket in buckets:
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        pending.append((work, flat, bucket))
    for work, flat, bucket in pending:
        work.wait()
        off = 0
        for p in bucket:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n


class GradientReducer:
    """SUM-all-reduce of parameter gradients overlapped with backward.

    Parameters are bucketed (~bucket_bytes of flat fp32 each) in REVERSE registration order -- the order
    their gradients appear in.  A post-accumulate-grad hook counts a bucket's gradients; the last one
    gathers them into the bucket's persistent flat buffer with one multi-tensor copy, starts the
    asynchronous all-reduce (RCCL runs it on its own stream, ordered after the copy) and re-points every
    `.grad` at its slice of the buffer, so nothing is copied back.  `finish()` -- call it between
    `backward()` and `optimizer.step()` -- reduces buckets that never filled (parameters without a gradient
    this step contribute zeros) and waits for all transfers.  Single-process: every method is a no-op.
    Every rank must build it over the same parameter list.
    """

    def __init__(self, params, group=None, bucket_bytes=32 << 20):
        self.group = group
        self.world = world_size(group)
        self.buckets = []
        self._where = {}
        self._handles = []
        if self.world == 1:
            return
        params = [p for p in params if p.requires_grad]
        cur, cur_bytes = [], 0
        groups = []
        for p in reversed(params):
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            groups.append(cur)
        for plist in groups:
            n = sum(p.numel() for p in plist)
            flat = torch.empty(n, dtype=plist[0].dtype, device=plist[0].device)
            views, off = [], 0
            for p in plist:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            bi = len(self.buckets)
            self.buckets.append({"params": plist, "flat": flat, "views": views, "ready": 0, "work": None, "events": []})
            for p in plist:
                self._where[p] = bi
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _on_grad(self, p):
        b = self.buckets[self._where[p]]
        if p.grad is not None and p.grad.is_cuda:
            # towers run on their own streams (models_multimodal.forward): the gradient exists on the stream this hook
            # runs under, which need not be the stream the bucket is finally gathered on
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(p.grad.device))
            b["events"].append(ev)
        b["ready"] += 1
        if b["ready"] == len(b["params"]) and b["work"] is None:
            self._launch(b)

    @torch.no_grad()
    def _launch(self, b):
        if b["events"]:
            cur = torch.cuda.current_stream(b["flat"].device)
            for ev in b["events"]:
                cur.wait_event(ev)
            b["events"] = []
        have = [(v, p.grad) for v, p in zip(b["views"], b["params"]) if p.grad is not None]
        if len(have) < len(b["params"]):
            b["flat"].zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        for v, p in zip(b["views"], b["params"]):
            p.grad = v

    def finish(self):
        for b in self.buckets:
            if b["work"] is None:
                self._launch(b)
        for b in self.buckets:
            b["work"].wait()
            b["work"], b["ready"], b["events"] = None, 0, []

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []

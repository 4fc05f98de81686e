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

# The pool's host driver only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails with
# "hipIpcGetMemHandle: invalid argument".  It has to be in the environment before the HIP runtime starts, i.e.
# before the first GPU call of the process -- importing this module early (the package does) is enough.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

from . import markers

# Set to a list to record every collective issued through this module (bench.py, tests): entries are
# (kind, payload bytes, start event | None, end event | None); None = no bookkeeping.  Events are recorded on the
# CALLER's stream around issue + wait, i.e. they measure how long the compute stream was held by the exchange.
COMM_LOG = None


class _Logged:
    __slots__ = ("entry",)

    def __init__(self, kind, t):
        self.entry = None
        if COMM_LOG is not None:
            ev0 = ev1 = None
            if t.is_cuda:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            self.entry = [kind, t.numel() * t.element_size(), ev0, ev1]

    def done(self):
        if self.entry is not None:
            if self.entry[3] is not None:
                self.entry[3].record()
            COMM_LOG.append(tuple(self.entry))


# Segmented HIP-graph capture of a data-parallel step (trainer.GraphedTrainStep): while a step is being recorded, every
# collective issued through this module ENDS the graph segment under capture and a new segment begins behind it; at replay the
# collective runs between the two graphs (the exchange is host-driven: gloo cannot be captured at all, and RCCL then needs no
# capture support either).  The object installed here implements exchange(fn): "close the segment, remember fn, open the next"
# -- fn is NOT run while recording (trainer._RecordedStep.exchange says why).
SEGMENTED_CAPTURE = None


def _exchange(fn):
    """Run the collective(s) of `fn` -- now, or as the break between two graph segments of a step being recorded."""
    if SEGMENTED_CAPTURE is not None:
        SEGMENTED_CAPTURE.exchange(fn)
    else:
        fn()


def all_gather_rows(t, group=None, kind="all_gather"):
    """All-gather equal-sized row blocks -> (world * b, ...) in rank order (no autograd)."""
    world = dist.get_world_size(group)
    t = t.contiguous()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)

    def run():
        log = _Logged(kind, out)
        with markers.range("exchange: " + kind):
            dist.all_gather_into_tensor(out, t, group=group)
        log.done()

    _exchange(run)
    return out


def all_reduce_sum(t, group=None, kind="all_reduce"):
    """In-place SUM all-reduce (no autograd)."""
    def run():
        log = _Logged(kind, t)
        with markers.range("exchange: " + kind):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        log.done()

    _exchange(run)
    return t


def backend_name(group=None):
    return dist.get_backend(group) if dist.is_available() and dist.is_initialized() else "none"


def init_from_env(backend=None):
    """Initialise from torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns
    (rank, local_rank, world_size).  A no-op for single-process runs (unless MSN_DIST_FORCE_INIT=1, which builds a
    one-rank group -- the RCCL smoke test on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        # more ranks than GPUs (a 1-GPU test box running the 2-rank flow over gloo) share devices round-robin
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    if (world > 1 or os.environ.get("MSN_DIST_FORCE_INIT") == "1") and not dist.is_initialized():
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

    def __init__(self, params, group=None, bucket_bytes=32 << 20, force=False, overlap=True):
        self.group = group
        self.overlap = overlap            # False: no autograd hooks, every bucket is reduced in finish() (graph-replayed steps)
        self.world = world_size(group)
        self.buckets = []
        self._where = {}
        self._handles = []
        if self.world == 1 and not (force and dist.is_available() and dist.is_initialized()):
            return                                   # force: run the machinery on a one-rank group (RCCL smoke test)
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
            self.buckets.append({"params": plist, "flat": flat, "views": views, "ready": 0, "work": None, "events": [],
                                 "stream_events": {}})
            for p in plist:
                self._where[p] = bi
                if overlap:
                    self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _on_grad(self, p):
        b = self.buckets[self._where[p]]
        if p.grad is not None and p.grad.is_cuda:
            # towers run on their own streams (models_multimodal.forward): the gradient exists on the stream this hook
            # runs under, which need not be the stream the bucket is finally gathered on.  One event per (bucket, stream)
            # -- re-recorded at every gradient of that stream, so it covers the last one -- instead of one per parameter
            # (150 event creations per step are ~0.7 ms of host time, which a 13-ms step at 128 rows per GPU does not have)
            s = torch.cuda.current_stream(p.grad.device)
            ev = b["stream_events"].get(s.cuda_stream)
            if ev is None:
                ev = b["stream_events"][s.cuda_stream] = torch.cuda.Event()
            ev.record(s)
            b["events"] = list(b["stream_events"].values())
        b["ready"] += 1
        if b["ready"] == len(b["params"]) and b["work"] is None:
            self._launch(b)

    @torch.no_grad()
    def _gather(self, b):
        """Copy a bucket's gradients into its flat buffer and re-point .grad at the slices (no collective)."""
        if b["events"]:
            cur = torch.cuda.current_stream(b["flat"].device)
            for ev in b["events"]:
                cur.wait_event(ev)
            b["events"] = []
        have = [(v, p.grad) for v, p in zip(b["views"], b["params"]) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        missing = any(p.grad is None for p in b["params"])
        if missing:
            for v, p in zip(b["views"], b["params"]):
                if p.grad is None:
                    v.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in zip(b["views"], b["params"]):
            p.grad = v

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
            if b["flat"].is_cuda:
                # the old .grad tensors are dropped below; one produced on another tower's side stream must not be
                # handed back to that stream's allocator pool while THIS stream's copy out of it is still queued
                cur = torch.cuda.current_stream(b["flat"].device)
                for _, g in have:
                    g.record_stream(cur)
        log = _Logged("grad_all_reduce", b["flat"])
        with markers.range("gradient all-reduce (bucket, issued under backward)"):
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        b["log"] = log
        for v, p in zip(b["views"], b["params"]):
            p.grad = v

    def finish(self):
        if not self.buckets:
            return
        if not self.overlap:
            # deferred form: gather every bucket, then all the all-reduces as ONE exchange (one break of a recorded step)
            for b in self.buckets:
                self._gather(b)

            def run():
                with markers.range("gradient all-reduce (all buckets)"):
                    for b in self.buckets:
                        log = _Logged("grad_all_reduce", b["flat"])
                        dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group)
                        log.done()

            _exchange(run)
            return
        for b in self.buckets:
            if b["work"] is None:
                self._launch(b)
        with markers.range("gradient all-reduce (wait)"):
            for b in self.buckets:
                b["work"].wait()
                b.pop("log").done()
                b["work"], b["ready"], b["events"] = None, 0, []

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []

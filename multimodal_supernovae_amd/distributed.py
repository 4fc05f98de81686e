"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" is
RCCL on ROCm) -- or gloo on CPU for the algebra tests.  The reference has no distributed code at
all; what is exchanged and why:

  * embeddings: all-gather of the local (b, D) rows per modality, then of the per-row LSEs
    (loss.py) -> every rank evaluates its own rows / columns of the global logit matrix;
  * parameter gradients: each rank's backward yields d(global loss)/d(params) through ITS samples,
    so the true gradient is the SUM over ranks (not the mean): bucketed all-reduce(SUM), launched
    asynchronously bucket by bucket so RCCL's transfers overlap each other on the xGMI links.
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
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
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

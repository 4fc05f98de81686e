#!/usr/bin/env python3
"""Two ranks sharing ONE GPU (gloo transport, CUDA tensors): full HIP training step with global negatives vs the
single-process step at the doubled batch.  A plumbing check for boxes with a single GPU; the real multi-GPU run
uses RCCL (backend "nccl") through the same code."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TK = dict(n_out=8, emb=16, heads=4, depth=2, dropout=0.0, time_norm=20583.37, agg="mean")
SK = dict(n_out=8, emb=8, heads=2, depth=2, dropout=0.0, time_norm=17945.14, agg="mean")


CK = dict(dim=8, depth=2, channels=3, kernel_size=5, patch_size=4, n_out=8, dropout_prob=0.0)
BATCHNORM = "--batchnorm" in sys.argv     # ConvMixer image tower + light curves, synchronised BatchNorm
TRAINER = "--trainer" in sys.argv         # Trainer.fit with a validation loader whose shards are UNEVEN across the ranks
GRAPHED = "--graphed" in sys.argv         # GraphedTrainStep under data parallel (segmented capture) == the eager steps, bit for bit
WORLD = int(sys.argv[sys.argv.index("--world") + 1]) if "--world" in sys.argv else 2   # default check only: ranks sharing the GPU (<= 4)


def make_model():
    from multimodal_supernovae_amd.models_multimodal import LightCurveImageCLIP
    torch.manual_seed(0)
    if BATCHNORM:
        return LightCurveImageCLIP(enc_dim=16, nband=2, transformer_kwargs=TK, conv_kwargs=CK,
                                   combinations=["host_galaxy", "lightcurve"], loss="softmax", lr=1e-2).cuda().train()
    return LightCurveImageCLIP(enc_dim=16, nband=2, transformer_kwargs=TK, transformer_spectral_kwargs=SK,
                               combinations=["lightcurve", "spectral"], loss="softmax", lr=1e-2).cuda().train()


def make_batch(n):
    g = torch.Generator().manual_seed(1)
    mask = torch.ones(n, 12, dtype=torch.bool)
    mask[:, 9:] = False
    if BATCHNORM:
        return (torch.rand(n, 3, 16, 16, generator=g), torch.randn(n, 12, generator=g), torch.rand(n, 12, generator=g) * 100,
                mask, None, None, None, None, None)
    return (None, torch.randn(n, 12, generator=g), torch.rand(n, 12, generator=g) * 100, mask,
            torch.randn(n, 10, generator=g), torch.rand(n, 10, generator=g) * 6000 + 3000,
            torch.ones(n, 10, dtype=torch.bool), None, None)


def trainer_worker(rank, world, port, out):
    """Two ranks through Trainer.fit: equal training shards (global negatives), validation shards of DIFFERENT length
    (rank 0: batches of 8 + 8 + 3 rows, rank 1: one batch of 5) -- the validation loop must not issue a collective per
    batch (it would hang or gather mismatched rows); the epoch's val_loss is the batch-weighted mean over both shards of
    each rank's local-negatives loss; an unequal TRAINING shard is refused up front."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from multimodal_supernovae_amd import distributed as D
    from multimodal_supernovae_amd.trainer import Trainer
    D.init_from_env(backend="gloo")
    full = make_batch(40)
    cut = lambda lo, hi: tuple(t[lo:hi] if t is not None else None for t in full)
    train = [cut(rank * 8, rank * 8 + 8), cut(16 + rank * 8, 16 + rank * 8 + 8)]
    val = [cut(0, 8), cut(8, 16), cut(16, 19)] if rank == 0 else [cut(32, 37)]
    model = make_model()
    tr = Trainer(max_epochs=1).fit(model, train, val)
    # single-process expectation of the validation number on the trained weights: local negatives per batch
    model.eval()
    model.global_negatives = False
    tot, rows = 0.0, 0
    with torch.no_grad():
        for b in [cut(0, 8), cut(8, 16), cut(16, 19), cut(32, 37)]:
            n = b[1].shape[0]
            tot += float(model._loss(model(*tuple(t.cuda() if t is not None else None for t in b)))) * n
            rows += n
    refused = False
    try:
        Trainer(max_epochs=1).fit(make_model(), train if rank == 0 else train[:1], None)
    except ValueError:
        refused = True
    out[f"r{rank}"] = (tr.history["val_loss"][-1], tot / rows, refused, tr.global_step)
    dist.barrier()
    dist.destroy_process_group()


def graphed_worker(rank, world, port, out):
    """Seven data-parallel steps (two eager warm-up steps, the recording, replays, ONE SHORT BATCH in between that runs
    eagerly) through trainer.GraphedTrainStep against the same seven steps issued eagerly with the hook-driven reducer:
    every loss and every parameter must be bit-identical on both ranks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from multimodal_supernovae_amd import distributed as D
    from multimodal_supernovae_amd.trainer import GraphedTrainStep
    D.init_from_env(backend="gloo")
    b, steps, short_at = 8, 7, 4
    full = make_batch(steps * world * b)
    batches = []
    for i in range(steps):
        lo = (i * world + rank) * b
        rows = 5 if i == short_at else b
        batches.append(tuple(t[lo:lo + rows].cuda() if t is not None else None for t in full))

    def run(graphed):
        model = make_model()
        D.broadcast_module(model)
        opt = model.configure_optimizers()["optimizer"]
        losses = []
        if graphed:
            step = GraphedTrainStep(model, opt, warmup=2)
            for i, batch in enumerate(batches):
                losses.append(float(step(batch, i).detach()))
            info = (step.graph.segments, step.graph.exchanges)
        else:
            reducer = D.GradientReducer(model.parameters(), bucket_bytes=64 << 10)
            for i, batch in enumerate(batches):
                opt.zero_grad(set_to_none=True)
                loss = model.training_step(batch, i)
                loss.backward()
                reducer.finish()
                opt.step()
                losses.append(float(loss.detach()))
            reducer.remove()
            info = None
        torch.cuda.synchronize()
        return [p.detach().clone() for p in model.parameters()], losses, info

    pe, le, _ = run(False)
    pg, lg, info = run(True)
    same_params = all(torch.equal(a, c) for a, c in zip(pe, pg))
    worst = max(float((a - c).abs().max()) for a, c in zip(pe, pg))
    out[f"r{rank}"] = (same_params, le == lg, worst, le, lg, info)
    dist.barrier()
    dist.destroy_process_group()


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from multimodal_supernovae_amd import distributed as D
    D.init_from_env(backend="gloo")
    if BATCHNORM:
        D.enable_sync_batchnorm()
    b = 8
    full = make_batch(world * b)
    local = tuple(t[rank * b:(rank + 1) * b].cuda() if t is not None else None for t in full)
    model = make_model()
    D.broadcast_module(model)
    reducer = D.GradientReducer(model.parameters(), bucket_bytes=64 << 10)   # several buckets, launched under backward
    loss = model.training_step(local, 0)
    loss.backward()
    reducer.finish()
    torch.cuda.synchronize()
    if rank == 0:
        dist.barrier()
        dist.destroy_process_group()       # single-process reference at the global batch
        D.enable_sync_batchnorm(enabled=False)
        ref = make_model()
        rl = ref.training_step(tuple(t.cuda() if t is not None else None for t in full), 0)
        rl.backward()
        worst, errs = 0.0, []
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            if k == "logit_bias":          # analytically zero gradient: rounding noise only
                continue
            scale = float(q.grad.abs().max()) + 1e-12
            e = float((p.grad - q.grad).abs().max()) / scale
            errs.append((e, k))
            worst = max(worst, e)
        out["top"] = sorted(errs)[-3:]
        out["loss"] = (float(loss.detach()), float(rl.detach()))
        out["worst_rel_grad_err"] = worst
        if BATCHNORM:                      # running statistics come from the global batch too
            sd, rd = model.state_dict(), ref.state_dict()
            out["worst_running_stat_err"] = max(float((sd[k] - rd[k]).abs().max()) for k in sd if "running_" in k)
    else:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    if GRAPHED:
        procs = [ctx.Process(target=graphed_worker, args=(r, 2, 29615, out)) for r in range(2)]
        [p.start() for p in procs]
        [p.join(300) for p in procs]
        print(dict(out), [p.exitcode for p in procs])
        ok = all(p.exitcode == 0 for p in procs) and len(out) == 2
        for same_params, same_losses, worst, le, lg, info in out.values():
            ok = ok and same_params and same_losses and len(le) == 7 and info[0] >= 4 and info[1] >= 4
        print("DIST CHECK", "OK" if ok else "FAILED")
        sys.exit(0 if ok else 1)
    if TRAINER:
        procs = [ctx.Process(target=trainer_worker, args=(r, 2, 29613, out)) for r in range(2)]
        [p.start() for p in procs]
        [p.join(300) for p in procs]
        print(dict(out), [p.exitcode for p in procs])
        ok = all(p.exitcode == 0 for p in procs) and len(out) == 2
        for got, want, refused, steps in out.values():
            ok = ok and abs(got - want) <= 1e-4 * abs(want) and refused and steps == 2
        ok = ok and abs(out["r0"][0] - out["r1"][0]) < 1e-9          # both ranks report the same global number
        print("DIST CHECK", "OK" if ok else "FAILED")
        sys.exit(0 if ok else 1)
    procs = [ctx.Process(target=worker, args=(r, WORLD, 29611, out)) for r in range(WORLD)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    print(dict(out), [p.exitcode for p in procs])
    ok = all(p.exitcode == 0 for p in procs) and abs(out["loss"][0] - out["loss"][1]) < 1e-4 * abs(out["loss"][1]) \
        and out["worst_rel_grad_err"] < 1e-3 and out.get("worst_running_stat_err", 0.0) < 1e-5
    print("DIST CHECK", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)

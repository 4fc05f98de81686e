#!/usr/bin/env python3
"""RCCL through the product's own exchange path on a ONE-GPU box: a one-rank "nccl" (= RCCL on ROCm) process group,
the sharded InfoNCE forced on (`global_negatives="always"`) and the gradient reducer forced on, so the packed embedding
all-gather, the LSE all-gather, the loss all-reduce and the bucketed gradient all-reduce all run as RCCL collectives
on device buffers.  With one rank every collective is the identity, so loss and gradients must equal the plain
single-process step.  The same one-rank RCCL group then carries the exchanges BETWEEN the graph segments of a replayed
step (trainer.GraphedTrainStep): five steps equal the eager ones bit for bit.  (Two ranks cannot share a device under RCCL; the 2-rank algebra is tested over gloo.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(MSN_DIST_FORCE_INIT="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                  MASTER_PORT=os.environ.get("MASTER_PORT", "29633"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from multimodal_supernovae_amd import distributed as D  # noqa: E402
from multimodal_supernovae_amd.models_multimodal import LightCurveImageCLIP  # noqa: E402

TK = dict(n_out=8, emb=16, heads=4, depth=2, dropout=0.0, time_norm=20583.37, agg="mean")
SK = dict(n_out=8, emb=8, heads=2, depth=2, dropout=0.0, time_norm=17945.14, agg="mean")
CK = dict(dim=8, depth=1, channels=3, kernel_size=5, patch_size=4, n_out=8, dropout_prob=0.0)


def make(global_negatives):
    torch.manual_seed(0)
    return LightCurveImageCLIP(enc_dim=16, nband=2, transformer_kwargs=TK, transformer_spectral_kwargs=SK, conv_kwargs=CK,
                               combinations=["host_galaxy", "lightcurve", "spectral"], loss="softmax", lr=1e-2,
                               global_negatives=global_negatives).cuda().train()


def main():
    rank, local, world = D.init_from_env(backend="nccl")
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    g = torch.Generator().manual_seed(1)
    n = 16
    batch = tuple(t.cuda() if t is not None else None for t in (
        torch.rand(n, 3, 16, 16, generator=g), torch.randn(n, 12, generator=g), torch.rand(n, 12, generator=g) * 100,
        torch.ones(n, 12, dtype=torch.bool), torch.randn(n, 10, generator=g),
        torch.rand(n, 10, generator=g) * 6000 + 3000, torch.ones(n, 10, dtype=torch.bool), None, None))
    ref = make(False)
    ref.training_step(batch, 0).backward()
    model = make("always")
    reducer = D.GradientReducer(model.parameters(), bucket_bytes=16 << 10, force=True)
    D.COMM_LOG = []
    loss = model.training_step(batch, 0)
    loss.backward()
    reducer.finish()
    torch.cuda.synchronize()
    kinds = [e[0] for e in D.COMM_LOG]
    ms = sum(e[2].elapsed_time(e[3]) for e in D.COMM_LOG)
    D.COMM_LOG = None
    ref_loss = float(ref.training_step(batch, 0).detach())
    worst = 0.0
    for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if k == "logit_bias":
            continue
        worst = max(worst, float((p.grad - q.grad).abs().max()) / (float(q.grad.abs().max()) + 1e-12))
    exchange = [k for k in kinds if k != "grad_all_reduce"]
    ok = (exchange == ["embedding_all_gather", "lse_all_gather", "loss_all_reduce"] and kinds.count("grad_all_reduce") >= 2
          and abs(float(loss.detach()) - ref_loss) <= 1e-6 * abs(ref_loss) and worst < 1e-5)
    print({"backend": dist.get_backend(), "collectives": kinds, "comm_ms": ms, "loss": float(loss.detach()), "ref_loss": ref_loss,
           "worst_rel_grad_err": worst})
    # ---- graph replay with RCCL carrying the exchanges between the graph segments (trainer.GraphedTrainStep): five steps
    # (two eager warm-up steps, the recording, replays) against the same five steps issued eagerly, bit for bit
    from multimodal_supernovae_amd.trainer import GraphedTrainStep

    def five_steps(graphed):
        m = make("always")
        opt = m.configure_optimizers()["optimizer"]
        red = D.GradientReducer(m.parameters(), bucket_bytes=16 << 10, force=True, overlap=not graphed)
        losses, info = [], None
        if graphed:
            step = GraphedTrainStep(m, opt, warmup=2, reducer=red)
            for i in range(5):
                losses.append(float(step(batch, i).detach()))
            info = (step.graph.segments, step.graph.exchanges)
        else:
            for i in range(5):
                opt.zero_grad(set_to_none=True)
                l = m.training_step(batch, i)
                l.backward()
                red.finish()
                opt.step()
                losses.append(float(l.detach()))
        red.remove()
        torch.cuda.synchronize()
        return [p.detach().clone() for p in m.parameters()], losses, info

    pe, le, _ = five_steps(False)
    pg, lg, info = five_steps(True)
    graphed_ok = le == lg and all(torch.equal(a, c) for a, c in zip(pe, pg)) and info[0] >= 4 and info[1] >= 4
    print({"graphed == eager": graphed_ok, "segments / exchanges": info, "losses": lg})
    ok = ok and graphed_ok
    dist.destroy_process_group()
    print("RCCL SMOKE", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

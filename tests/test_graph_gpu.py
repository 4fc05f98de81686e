"""The training step recorded as a HIP graph (trainer.GraphedTrainStep: zero_grad -> training_step -> backward -> fused RAdam
with its step count on the device) against the same steps run eagerly: identical loss trajectory and parameters."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

TK = dict(n_out=8, emb=16, heads=4, depth=2, dropout=0.0, time_norm=20583.37, agg="mean")
SK = dict(n_out=8, emb=8, heads=2, depth=2, dropout=0.0, time_norm=17945.14, agg="mean")
CK = dict(dim=8, depth=2, channels=3, kernel_size=5, patch_size=4, n_out=8, dropout_prob=0.0)


def _model(combos):
    from multimodal_supernovae_amd.models_multimodal import LightCurveImageCLIP
    torch.manual_seed(0)
    return LightCurveImageCLIP(enc_dim=16, nband=2, transformer_kwargs=TK, transformer_spectral_kwargs=SK, conv_kwargs=CK,
                               combinations=combos, loss="softmax", lr=3e-3,
                               optimizer_kwargs={"weight_decay": 1e-3}).cuda().train()


def _batches(n, combos, steps):
    g = torch.Generator().manual_seed(5)
    out = []
    for _ in range(steps):
        mask = torch.ones(n, 12, dtype=torch.bool)
        mask[:, 9:] = torch.rand(n, 3, generator=g) > 0.5
        img = torch.rand(n, 3, 16, 16, generator=g) if "host_galaxy" in combos else None
        sp = (torch.randn(n, 10, generator=g), torch.rand(n, 10, generator=g) * 6000 + 3000,
              torch.ones(n, 10, dtype=torch.bool)) if "spectral" in combos else (None, None, None)
        b = (img, torch.randn(n, 12, generator=g), torch.rand(n, 12, generator=g) * 100, mask, *sp, None, None)
        out.append(tuple(t.cuda() if t is not None else None for t in b))
    return out


@pytest.mark.parametrize("concurrent", [True, False])
@pytest.mark.parametrize("combos", [["lightcurve", "spectral"], ["host_galaxy", "lightcurve"]])
def test_graphed_step_equals_eager_steps(combos, concurrent):
    from multimodal_supernovae_amd.trainer import GraphedTrainStep
    steps = 9                                           # crosses RAdam's rectification switch (rho_t > 5 from step 6)
    batches = _batches(8, combos, steps)
    eager = _model(combos)
    graphed = copy.deepcopy(eager)
    opt_e = eager.configure_optimizers()["optimizer"]
    losses_e = []
    for b in batches:
        opt_e.zero_grad(set_to_none=True)
        loss = eager.training_step(b, 0)
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss.detach()))
    opt_g = graphed.configure_optimizers()["optimizer"]
    step = GraphedTrainStep(graphed, opt_g, warmup=3, concurrent_towers=concurrent)
    losses_g = [float(step(b).detach()) for b in batches]
    assert step.graph is not None and step.calls == steps
    torch.cuda.synchronize()
    for a, b in zip(losses_e, losses_g):
        assert abs(a - b) <= 1e-5 * abs(a), (losses_e, losses_g)
    for (k, p), (_, q) in zip(eager.named_parameters(), graphed.named_parameters()):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-7, msg=lambda m: f"{k}: {m}")
    for (k, p), (_, q) in zip(eager.named_buffers(), graphed.named_buffers()):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-7, msg=lambda m: f"buffer {k}: {m}")
    assert all(int(st["step"]) == steps for st in opt_g.state.values())


def test_graphed_step_with_an_odd_batch_in_between():
    """A batch of another shape (the short last batch of an epoch) runs eagerly; the device-side step count follows."""
    from multimodal_supernovae_amd.trainer import GraphedTrainStep
    combos = ["lightcurve", "spectral"]
    batches = _batches(8, combos, 8)
    batches[5] = tuple(t[:5] if t is not None else None for t in batches[5])
    eager = _model(combos)
    graphed = copy.deepcopy(eager)
    opt_e = eager.configure_optimizers()["optimizer"]
    for b in batches:
        opt_e.zero_grad(set_to_none=True)
        eager.training_step(b, 0).backward()
        opt_e.step()
    step = GraphedTrainStep(graphed, graphed.configure_optimizers()["optimizer"], warmup=2)
    for b in batches:
        step(b)
    torch.cuda.synchronize()
    for (k, p), (_, q) in zip(eager.named_parameters(), graphed.named_parameters()):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-7, msg=lambda m: f"{k}: {m}")


def test_trainer_with_graphed_steps_matches_eager_trainer():
    """Trainer(graphed_steps=True) over two epochs whose last batch is short == the eager Trainer (losses, parameters)."""
    from multimodal_supernovae_amd.trainer import Trainer
    combos = ["lightcurve", "spectral"]
    batches = _batches(8, combos, 6)
    batches[-1] = tuple(t[:3] if t is not None else None for t in batches[-1])
    cpu = [tuple(t.cpu() if t is not None else None for t in b) for b in batches]
    a, b = _model(combos), None
    b = copy.deepcopy(a)
    ta = Trainer(max_epochs=2).fit(a, cpu)
    tb = Trainer(max_epochs=2, graphed_steps=True).fit(b, cpu)
    torch.cuda.synchronize()
    for x, y in zip(ta.history["train_loss"], tb.history["train_loss"]):
        assert abs(x - y) <= 1e-5 * abs(x)
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-7, msg=lambda m: f"{k}: {m}")


def test_graphed_step_with_dropout_draws_new_masks_and_is_reproducible():
    """Dropout inside the recorded step: device-resident seeds (a base advanced once per replay + the call's ordinal).
    Replays on ONE batch at lr = 0 give different losses (new masks each time), and the whole sequence repeats exactly
    under the same torch seed."""
    from multimodal_supernovae_amd.models_multimodal import LightCurveImageCLIP
    from multimodal_supernovae_amd.trainer import GraphedTrainStep
    tk = dict(TK, dropout=0.3)
    sk = dict(SK, dropout=0.3)
    combos = ["lightcurve", "spectral"]
    batch = _batches(8, combos, 1)[0]

    def run():
        torch.manual_seed(11)
        m = LightCurveImageCLIP(enc_dim=16, nband=2, transformer_kwargs=tk, transformer_spectral_kwargs=sk, conv_kwargs=CK,
                                combinations=combos, loss="softmax", lr=0.0).cuda().train()
        step = GraphedTrainStep(m, m.configure_optimizers()["optimizer"], warmup=2)
        out = [float(step(batch).detach()) for _ in range(7)]
        assert step.graph is not None
        g = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        return out, g

    a, ga = run()
    b, gb = run()
    assert a == b                                        # reproducible, replays included
    replays = a[2:]
    assert len({round(v, 6) for v in replays}) >= 5, replays      # new masks at every replay
    assert all(torch.isfinite(torch.tensor(a)))
    for k in ga:
        assert torch.equal(ga[k], gb[k]) and torch.isfinite(ga[k]).all(), k

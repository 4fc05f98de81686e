"""GPU parity of the image tower, the MLP, the full CLIP module and the training harness against
golden vectors from the reference (tools/gen_golden.py).  1e-3 relative, magnitude-scaled."""
import pytest
import torch

from conftest import Fixture, golden_names

pytestmark = pytest.mark.gpu
RTOL = 1e-3
BATCH_KEYS = ["x_img", "x_lc", "t_lc", "mask_lc", "x_sp", "t_sp", "mask_sp", "redshift", "classification"]


def close(a, b, what, rtol=RTOL):
    b = b.to(a.dtype)
    scale = float(b.abs().max()) + 1e-6
    torch.testing.assert_close(a, b, rtol=rtol, atol=rtol * scale * 0.1, msg=lambda m: f"{what}: {m}")


def _batch(ins, prefix="", device="cuda"):
    return tuple(ins[prefix + k].to(device) if prefix + k in ins else None for k in BATCH_KEYS)


@pytest.mark.parametrize("name", golden_names("convmixer_"))
def test_convmixer(name):
    from multimodal_supernovae_amd.models_multimodal import ConvMixer
    f = Fixture(name)
    c = f.cfg
    m = ConvMixer(dim=c["dim"], depth=c["depth"], channels=c["channels"], kernel_size=c["kernel_size"],
                  patch_size=c["patch_size"], n_out=c["n_out"], dropout_prob=0.0)
    m.load_state_dict(f.P, strict=True)
    m.cuda().train(c["mode"] == "train")
    x = f.groups["in"]["x"].cuda().requires_grad_()
    y = m(x)
    close(y.detach().cpu(), f.out["y"], "y")
    y.backward(f.groups["in"]["cot"].cuda())
    close(x.grad.cpu(), f.grad["x"], "dx")
    for k, p in m.named_parameters():
        close(p.grad.cpu(), f.grad[k], "grad " + k)
    sd = m.state_dict()
    for k, v in f.stats.items():   # running statistics after the step (unchanged in eval mode)
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v), k
        else:
            close(sd[k].cpu(), v, k, rtol=1e-4)


def test_mlp():
    from multimodal_supernovae_amd.models_multimodal import MLP
    f = Fixture("mlp")
    m = MLP(input_dim=12, hidden_dim=16, output_dim=8, num_layers=2, dropout=0.0)
    m.load_state_dict(f.P, strict=True)
    m.cuda()
    x = f.groups["in"]["x"].cuda().requires_grad_()
    y = m(x)
    close(y.detach().cpu(), f.out["y"], "y")
    y.backward(f.groups["in"]["cot"].cuda())
    close(x.grad.cpu(), f.grad["x"], "dx")
    for k, p in m.named_parameters():
        close(p.grad.cpu(), f.grad[k], "grad " + k)


def _build(cfg):
    from multimodal_supernovae_amd.models_multimodal import LightCurveImageCLIP
    return LightCurveImageCLIP(enc_dim=cfg["enc_dim"], logit_scale=10.0, nband=cfg["nband"],
                               transformer_kwargs=cfg["transformer_kwargs"],
                               transformer_spectral_kwargs=cfg["transformer_spectral_kwargs"],
                               conv_kwargs=cfg["conv_kwargs"], meta_kwargs=cfg["meta_kwargs"],
                               combinations=cfg["combinations"], optimizer_kwargs={"weight_decay": cfg["weight_decay"]},
                               lr=cfg["lr"], loss=cfg["loss"])


@pytest.mark.parametrize("name", golden_names("clip_"))
def test_clip_module_training_step(name):
    f = Fixture(name)
    model = _build(f.cfg)
    assert set(model.state_dict().keys()) == set(f.P.keys())
    model.load_state_dict(f.P, strict=True)
    model.cuda().train()
    batch = _batch(f.groups["in"])
    embs = model(*batch)
    assert len(embs) == len(f.cfg["combinations"])
    for k, e in enumerate(embs):                       # fixed order img, lc, sp, meta
        close(e.detach().cpu(), f.out[f"emb{k}"], f"emb{k}")
    model.zero_grad()
    model = _build(f.cfg)                              # fresh BN running stats for the step itself
    model.load_state_dict(f.P, strict=True)
    model.cuda().train()
    loss = model.training_step(batch, 0)
    assert abs(float(loss.detach()) - float(f.out["loss"])) <= RTOL * abs(float(f.out["loss"]))
    loss.backward()
    for k, p in model.named_parameters():
        assert k in f.grad, k
        if k == "logit_bias" and f.cfg["loss"] == "softmax":      # analytically zero (the bias cancels in both log-softmaxes): rounding noise only
            assert abs(float(p.grad)) < 1e-5 and abs(float(f.grad[k])) < 1e-5
            continue
        close(p.grad.cpu(), f.grad[k], "grad " + k, rtol=2e-3)


def test_harness_eight_radam_steps_match_reference():
    """the reference's loop (zero_grad -> training_step -> backward -> RAdam.step) over two
    cycled batches for 8 steps (crossing RAdam's rectification switch): loss trajectory + final weights."""
    from multimodal_supernovae_amd.trainer import Trainer
    f = Fixture("harness_radam")
    model = _build(f.cfg)
    model.load_state_dict(f.P, strict=True)
    batches = [_batch(f.groups["in"], f"b{i}.", device="cpu") for i in range(2)]
    tr = Trainer(max_epochs=f.cfg["n_steps"] // 2).fit(model, batches)
    got = torch.stack(tr.step_losses).cpu()
    torch.testing.assert_close(got, f.out["losses"], rtol=2e-3, atol=1e-4)
    sd = model.state_dict()
    for k, v in f.after.items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v)
        else:
            close(sd[k].cpu(), v, "after " + k, rtol=5e-3)


def test_radam_kernel_matches_torch_optimizer():
    from multimodal_supernovae_amd.optim import RAdam
    g = torch.Generator().manual_seed(3)
    shapes = [(7, 3), (5,), (), (129, 33), (1000,)]
    w0 = [torch.randn(s, generator=g) for s in shapes]
    a = [w.clone().cuda().requires_grad_() for w in w0]
    b = [w.clone().requires_grad_() for w in w0]
    oa, ob = RAdam(a, lr=1e-2, weight_decay=1e-3), torch.optim.RAdam(b, lr=1e-2, weight_decay=1e-3)
    for step in range(10):
        for ps, opt in ((a, oa), (b, ob)):
            opt.zero_grad()
            for p in ps:
                p.grad = (torch.sin(p.detach() * (step + 1)) + 0.1).to(p.device)
            opt.step()
        for pa, pb in zip(a, b):
            torch.testing.assert_close(pa.detach().cpu(), pb.detach(), rtol=1e-5, atol=1e-6)
    assert set(oa.state[a[0]].keys()) == {"step", "exp_avg", "exp_avg_sq"}


def test_cpu_module_fails_loudly():
    """No CPU fallback: calling the product path without the GPU raises instead of computing."""
    from multimodal_supernovae_amd import _lib
    from multimodal_supernovae_amd.models_multimodal import MLP
    m = MLP(input_dim=4, hidden_dim=4, output_dim=2, num_layers=1, dropout=0.0)
    with pytest.raises(_lib.MsnHipError):
        m(torch.randn(3, 4))


def test_real_reference_checkpoint_loads_strict_and_matches():
    """Row f4: a Lightning-2.2.3 checkpoint shipped with the reference loads strict=True into the HIP-backed
    module; embeddings / loss on a fixed synthetic lc + spectrum batch (T = 200 / 220, Maven sizes) match."""
    f = Fixture("real_ckpt_lc_sp")
    c = f.cfg
    from multimodal_supernovae_amd.models_multimodal import LightCurveImageCLIP
    model = LightCurveImageCLIP(enc_dim=c["enc_dim"], logit_scale=19.545966923442453, nband=c["nband"],
                                transformer_kwargs=c["transformer_kwargs"],
                                transformer_spectral_kwargs=c["transformer_spectral_kwargs"],
                                combinations=c["combinations"], loss="softmax")
    model.load_state_dict(f.P, strict=True)
    model.cuda().eval()
    batch = _batch(f.groups["in"])
    with torch.no_grad():
        embs = model(*batch)
        loss = model.validation_step(batch, 0) if model.embs_list is not None else model._loss(embs)
    for k, e in enumerate(embs):
        close(e.cpu(), f.out[f"emb{k}"], f"emb{k}")
    assert abs(float(loss) - float(f.out["loss"])) <= RTOL * abs(float(f.out["loss"]))


def test_masked_pretraining_matches_reference():
    """Row f4: MaskedLightCurveEncoder forward, masked MSE and parameter gradients vs the reference golden."""
    from multimodal_supernovae_amd.models_pretraining import MaskedLightCurveEncoder, get_continous_random_mask
    f = Fixture("pretraining")
    c, i = f.cfg, f.groups["in"]
    m = MaskedLightCurveEncoder(f_mask=c["f_mask"], nband=c["nband"], transformer_kwargs=c["transformer_kwargs"])
    assert set(m.state_dict().keys()) == set(f.P.keys())
    m.load_state_dict(f.P, strict=True)
    m.cuda().train()
    x, t, pad = i["x"].cuda(), i["t"].cuda(), i["padding_mask"].cuda()
    xm = x.clone()
    xm[~i["mask_in"].cuda()] = 0
    pred = m(xm, t, mask=pad)
    close(pred.detach().cpu(), f.out["pred"], "pred")
    loss = m.masked_loss(x, t, pad, i["mask_in"].cuda(), i["mask_pred"].cuda())
    assert abs(float(loss.detach()) - float(f.out["loss"])) <= RTOL * abs(float(f.out["loss"]))
    loss.backward()
    for k, p in m.named_parameters():
        if k in ("net.projection.weight", "net.projection.bias"):       # unused by agg="pretraining"
            continue
        close(p.grad.cpu(), f.grad[k], "grad " + k, rtol=2e-3)
    # the training hook draws its own contiguous masks: one hidden run per band, inside the observed points
    mask_in, mask_pred = get_continous_random_mask(i["padding_mask"], c["nband"], f_mask=c["f_mask"])
    assert not (mask_pred & ~i["padding_mask"]).any() and not (mask_in & mask_pred).any()
    assert torch.isfinite(m.training_step((t, x, pad), 0).detach())


@pytest.mark.parametrize("B", [1, 2, 3, 7])
def test_small_and_odd_batches_match_oracle(B):
    """Edge sizes of the step itself: a single pair (the InfoNCE of one pair is 0 with gradient 0 to the
    embeddings), odd batches, ragged masks down to one valid time step -- HIP module against the oracle."""
    from oracle import clip as oclip
    cfg = dict(enc_dim=16, nband=2, combinations=["lightcurve", "spectral"],
               transformer_kwargs=dict(n_out=8, emb=16, heads=2, depth=2, dropout=0.0, time_norm=1000.0, agg="mean"),
               transformer_spectral_kwargs=dict(n_out=8, emb=16, heads=4, depth=1, dropout=0.0, time_norm=5000.0, agg="max"),
               conv_kwargs=dict(dim=8, depth=1, channels=3, kernel_size=5, patch_size=8, n_out=8, dropout_prob=0.0),
               meta_kwargs=None, weight_decay=0.0, lr=1e-3, loss="softmax")
    torch.manual_seed(40 + B)
    model = _build(cfg)
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(B)
    T_lc, T_sp = 10, 9
    mask_lc = torch.arange(T_lc)[None, :] < torch.randint(1, T_lc + 1, (B, 1), generator=g)
    mask_sp = torch.arange(T_sp)[None, :] < torch.randint(1, T_sp + 1, (B, 1), generator=g)
    batch = (None, torch.randn(B, T_lc, generator=g), torch.rand(B, T_lc, generator=g) * 100, mask_lc,
             torch.randn(B, T_sp, generator=g), torch.rand(B, T_sp, generator=g) * 6000 + 3000, mask_sp, None, None)
    ref = oclip.training_loss(P, cfg, batch, loss="softmax")
    ref.backward()
    model.cuda().train()
    loss = model.training_step(tuple(t.cuda() if torch.is_tensor(t) else t for t in batch), 0)
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-5 + RTOL * abs(float(ref.detach()))
    loss.backward()
    for k, p in model.named_parameters():
        want = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(want)
        if k == "logit_bias":
            assert abs(float(got)) < 1e-5
            continue
        scale = float(want.abs().max()) + 1e-6
        assert float((got - want).abs().max()) <= 2e-3 * scale + 1e-6, (k, B)

"""The oracle (oracle/) against golden vectors produced by the real reference
(tools/gen_golden.py).  CPU only.  Tolerances: fp32, 2e-5 relative / 1e-6 absolute --
the restatement uses different (but algebraically identical) op sequences."""
import math

import pytest
import torch

from conftest import Fixture, golden_names
from oracle import clip as oclip
from oracle import encoders as oenc
from oracle import loss as oloss

RTOL, ATOL = 2e-5, 2e-6


def close(a, b, rtol=RTOL, atol=ATOL, what=""):
    torch.testing.assert_close(a, b.to(a.dtype), rtol=rtol, atol=atol, msg=lambda m: f"{what}: {m}")


def test_loss_known_answers():
    f = Fixture("loss_kat")
    eye = torch.eye(4)
    close(oloss.clip_loss(eye, eye, torch.tensor(0.0), torch.tensor(0.0)), f.out["clip_I4_s0_b0"])
    assert abs(float(f.out["clip_I4_s0_b0"]) - math.log(1 + 3 / math.e)) < 1e-6
    close(oloss.clip_loss(eye, eye, torch.tensor(math.log(10.0)), torch.tensor(-10.0)),
          f.out["clip_I4_ln10_bm10"], rtol=1e-4)
    close(oloss.sigmoid_loss(eye, eye, torch.tensor(math.log(10.0)), torch.tensor(-10.0)),
          f.out["sigmoid_I4_ln10_bm10"])
    assert abs(float(f.out["sigmoid_I4_ln10_bm10"]) - 7.5) < 1e-3


@pytest.mark.parametrize("name", golden_names("loss_clip_n") + golden_names("loss_sigmoid_n"))
def test_pair_loss_and_grads(name):
    f = Fixture(name)
    ins = {k: v.clone().requires_grad_() for k, v in f.groups["in"].items()}
    fn = oloss.clip_loss if "clip" in name else oloss.sigmoid_loss
    loss = fn(ins["e1"], ins["e2"], ins["logit_scale"], ins["logit_bias"])
    close(loss, f.out["loss"], what="loss")
    loss.backward()
    for k in ["e1", "e2", "logit_scale", "logit_bias"]:
        close(ins[k].grad, f.grad[k], atol=1e-7 if k.startswith("e") else 2e-6, what=k)
    if "clip" in name:  # the bias shifts every logit: exactly gradient-free
        assert abs(float(f.grad["logit_bias"])) < 1e-6


def test_clip_loss_unequal_lengths():
    f = Fixture("loss_clip_unequal")
    i = f.groups["in"]
    close(oloss.clip_loss(i["e1"], i["e2"], i["logit_scale"], i["logit_bias"]), f.out["loss"])


def test_clip_loss_multimodal_three_way():
    f = Fixture("loss_clip_multimodal3")
    ins = {k: v.clone().requires_grad_() if v.is_floating_point() else v for k, v in f.groups["in"].items()}
    embs = [ins["e0"], ins["e1"], ins["e2"]]
    loss = oloss.clip_loss_multimodal(embs, ins["logit_scale"], ins["logit_bias"])
    close(loss, f.out["loss"])
    loss.backward()
    for k in ["e0", "e1", "e2", "logit_scale", "logit_bias"]:
        close(ins[k].grad, f.grad[k], what=k)
    det = [e.detach() for e in embs]
    close(oloss.clip_loss_multimodal(det, f.groups["in"]["scales_vec"], f.groups["in"]["biases_vec"]),
          f.out["loss_vec"])
    close(oloss.sigmoid_loss_multimodal(det, ins["logit_scale"].detach(), ins["logit_bias"].detach()),
          f.out["sigmoid"])


def _check_grads(P, f, skip=()):
    for k, g in f.grad.items():
        if k in skip or k not in P:
            continue
        assert P[k].grad is not None, k
        close(P[k].grad, g, rtol=2e-4, atol=2e-5, what="grad " + k)


@pytest.mark.parametrize("name", golden_names("attn_") + golden_names("block_"))
def test_attention_and_block(name):
    f = Fixture(name)
    P = f.params()
    x = f.groups["in"]["x"].clone().requires_grad_()
    fn = oenc.self_attention if name.startswith("attn") else oenc.transformer_block
    y = fn(P, "", x, f.groups["in"]["mask"], f.cfg["heads"])
    close(y, f.out["y"], rtol=1e-4, atol=1e-5)
    (y * f.groups["in"]["cot"]).sum().backward()
    close(x.grad, f.grad["x"], rtol=2e-4, atol=2e-5)
    _check_grads(P, f)


def test_time_positional_encoding():
    f = Fixture("timeenc")
    close(oenc.time_positional_encoding(f.groups["in"]["t"], f.cfg["emb"], f.cfg["norm"]), f.out["pe"],
          rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", golden_names("tenc_"))
def test_transformer_with_time_embeddings(name):
    f = Fixture(name)
    P = f.params()
    i, c = f.groups["in"], f.cfg
    y = oenc.transformer_with_time_embeddings(P, "", i["x"], i["t"], i["mask"], emb=c["emb"], heads=c["heads"],
                                              depth=c["depth"], time_norm=c["time_norm"], nband=c["nband"],
                                              agg=c["agg"])
    close(y, f.out["y"], rtol=1e-4, atol=1e-5)
    (y * i["cot"]).sum().backward()
    _check_grads(P, f)


@pytest.mark.parametrize("name", golden_names("convmixer_"))
def test_convmixer(name):
    f = Fixture(name)
    P = f.params()
    x = f.groups["in"]["x"].clone().requires_grad_()
    stats = {}
    y = oenc.convmixer(P, "", x, depth=f.cfg["depth"], patch_size=f.cfg["patch_size"],
                       training=f.cfg["mode"] == "train", stats_out=stats)
    close(y, f.out["y"], rtol=1e-4, atol=1e-5)
    (y * f.groups["in"]["cot"]).sum().backward()
    close(x.grad, f.grad["x"], rtol=5e-4, atol=2e-5)
    _check_grads(P, f)
    if f.cfg["mode"] == "train":
        for k, v in stats.items():
            close(v, f.stats[k], rtol=1e-5, atol=1e-6, what=k)
    else:
        assert not stats


def test_mlp():
    f = Fixture("mlp")
    P = f.params()
    x = f.groups["in"]["x"].clone().requires_grad_()
    y = oenc.mlp(P, "", x, f.cfg["num_layers"])
    close(y, f.out["y"])
    (y * f.groups["in"]["cot"]).sum().backward()
    close(x.grad, f.grad["x"])
    _check_grads(P, f)


def _batch(ins, prefix=""):
    keys = ["x_img", "x_lc", "t_lc", "mask_lc", "x_sp", "t_sp", "mask_sp", "redshift", "classification"]
    return tuple(ins.get(prefix + k) for k in keys)


@pytest.mark.parametrize("name", golden_names("clip_"))
def test_clip_training_step(name):
    f = Fixture(name)
    P = f.params()
    batch = _batch(f.groups["in"])
    embs = oclip.embeddings(P, f.cfg, batch, training=True)
    for k, e in enumerate(embs):
        close(e, f.out[f"emb{k}"], rtol=1e-4, atol=1e-5, what=f"emb{k}")
        close(e.norm(dim=-1), torch.ones(e.shape[0]), rtol=1e-5, atol=1e-5)
    loss = oclip.training_loss(P, f.cfg, batch, loss=f.cfg["loss"])
    close(loss, f.out["loss"], rtol=1e-4, atol=1e-5)
    loss.backward()
    for k, g in f.grad.items():
        close(P[k].grad, g, rtol=2e-3, atol=5e-5, what="grad " + k)


def test_radam_restatement_matches_torch():
    g = torch.Generator().manual_seed(5)
    w0 = [torch.randn(7, 3, generator=g), torch.randn(5, generator=g), torch.tensor(0.3)]
    a = [w.clone().requires_grad_() for w in w0]
    b = [w.clone().requires_grad_() for w in w0]
    opt_a = torch.optim.RAdam(a, lr=1e-2, weight_decay=1e-3)
    opt_b = oclip.RAdam(b, lr=1e-2, weight_decay=1e-3)
    for step in range(10):
        for ps, opt in ((a, opt_a), (b, opt_b)):
            opt.zero_grad()
            loss = sum(((p * (1 + 0.1 * step)) ** 2).sum() + p.sum() for p in ps)
            loss.backward()
            opt.step()
        for pa, pb in zip(a, b):
            close(pb.detach(), pa.detach(), rtol=1e-6, atol=1e-7, what=f"step {step}")


def test_harness_loss_trajectory_and_final_weights():
    """zero_grad -> training_step -> backward -> RAdam.step, 8 steps, 2 batches cycled."""
    f = Fixture("harness_radam")
    P = f.params()
    batches = [_batch(f.groups["in"], f"b{i}.") for i in range(2)]
    trainable = [v for v in P.values() if v.requires_grad]
    opt = oclip.RAdam(trainable, lr=f.cfg["lr"], weight_decay=f.cfg["weight_decay"])
    for step in range(f.cfg["n_steps"]):
        opt.zero_grad()
        stats = {}
        loss = oclip.training_loss(P, f.cfg, batches[step % 2], stats_out=stats)
        loss.backward()
        opt.step()
        with torch.no_grad():
            for k, v in stats.items():
                P[k].copy_(v)
        close(loss.detach(), f.out["losses"][step], rtol=2e-4, atol=2e-5, what=f"loss step {step}")
    for k, v in f.after.items():
        if k.endswith("num_batches_tracked"):
            continue
        close(P[k].detach(), v, rtol=2e-3, atol=2e-4, what="after " + k)


def test_real_reference_checkpoint_embeddings():
    """A checkpoint shipped with the reference (real trained weights): embeddings and loss on a fixed synthetic batch."""
    f = Fixture("real_ckpt_lc_sp")
    P = f.params(requires_grad=False)
    batch = _batch(f.groups["in"])
    embs = oclip.embeddings(P, f.cfg, batch, training=False)
    for k, e in enumerate(embs):
        close(e, f.out[f"emb{k}"], rtol=1e-4, atol=2e-5, what=f"emb{k}")
    close(oclip.training_loss(P, f.cfg, batch, training=False), f.out["loss"], rtol=1e-4, atol=1e-5)
    assert abs(float(P["logit_scale"].exp()) - 31.02) < 0.01


def test_retrieval_auc_matches_reference():
    import numpy as np
    f = Fixture("auc")
    for n in (50, 137):
        t, frac, _ = oclip.roc_data(f.groups["in"][f"e1_{n}"], f.groups["in"][f"e2_{n}"])
        assert np.allclose(t, f.out[f"thresholds_{n}"].numpy())
        assert np.allclose(frac, f.out[f"fraction_{n}"].numpy())
        assert abs(oclip.auc(f.groups["in"][f"e1_{n}"], f.groups["in"][f"e2_{n}"]) - float(f.out[f"auc_{n}"])) < 1e-12


def test_masked_pretraining_objective():
    """Row f4: MaskedLightCurveEncoder = transformer (agg="pretraining") + Linear(emb, 1); MSE on the hidden points."""
    f = Fixture("pretraining")
    P = f.params()
    i, c = f.groups["in"], f.cfg
    tk = c["transformer_kwargs"]
    xm = i["x"].clone()
    xm[~i["mask_in"]] = 0
    h = oenc.transformer_with_time_embeddings(P, "net.", xm[..., None], i["t"], i["padding_mask"], emb=tk["emb"],
                                              heads=tk["heads"], depth=tk["depth"], time_norm=tk["time_norm"],
                                              nband=c["nband"], agg="pretraining")
    pred = oenc.linear(P, "last_layer", h).squeeze(2)
    close(pred, f.out["pred"], rtol=1e-4, atol=1e-5)
    loss = ((pred - i["x"]) ** 2)[i["mask_pred"]].mean()
    close(loss, f.out["loss"], rtol=1e-4, atol=1e-6)
    loss.backward()
    for k, g in f.grad.items():
        close(P[k].grad, g, rtol=2e-3, atol=2e-5, what="grad " + k)


@pytest.mark.parametrize("name", ["val_loop_lc_sp", "val_loop_3tower"])
def test_validation_loop_values(name):
    """Row a15: per-batch val_loss and the retrieval AUC the reference logs from on_validation_epoch_end
    (src/models_multimodal.py:415-556), three validation batches of 6, 6 and 4 rows, eval mode."""
    f = Fixture(name)
    P = f.params(requires_grad=False)
    batches = [_batch(f.groups["in"], f"b{i}.") for i in range(3)]
    embs_list = None
    for i, batch in enumerate(batches):
        embs = oclip.embeddings(P, f.cfg, batch, training=False)
        embs_list = [[e] for e in embs] if embs_list is None else [acc + [e] for acc, e in zip(embs_list, embs)]
        loss = oclip.training_loss(P, f.cfg, batch, training=False)
        close(loss.double(), f.out["val_losses"][i], rtol=1e-4, atol=1e-5, what=f"val_loss batch {i}")
    cat = [torch.cat(e, dim=0) for e in embs_list]
    if len(cat) == 2:
        assert abs(oclip.auc(cat[0], cat[1]) - float(f.out["AUC_val"])) < 1e-12
    else:
        count = 1
        for i in range(len(cat) - 1):
            for j in range(i + 1, len(cat)):
                assert abs(oclip.auc(cat[i], cat[j]) - float(f.out[f"AUC_val{count}"])) < 1e-12, count
                count += 1
        assert count == 4
    assert f.cfg["logged_keys"][:3] == ["val_loss"] * 3


@pytest.mark.parametrize("name", golden_names("augment_"))
def test_series_augmentation_matches_reference_loader(name):
    """oracle.augment.series_noise on the field the REAL NoisyDataLoader drew == the batch it yielded (bit for bit)."""
    from oracle import augment as oaug
    fx = Fixture(name)
    checked = 0
    for key, want in fx.out.items():
        b, what = key.split(".")
        kind = what.split("_")[1]
        got = oaug.series_noise(fx.groups["in"][f"{b}.x_{kind}"], fx.groups["in"][f"{b}.err_{kind}"],
                                fx.groups["in"][f"{b}.field_{kind}"], fx.cfg["noise_level_mag"])
        assert torch.equal(got, want), key
        checked += 1
    assert checked == fx.cfg["batches"] * len(fx.cfg["combinations"])

#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference in the build container.

Runs only where /root/reference exists (never on the GPU box, never from tests).
It imports the reference's own modules -- src/loss.py and src/transformer_utils.py
directly, src/models_multimodal.py after registering inert stand-ins for the
third-party packages this image lacks (pytorch_lightning, ruamel.yaml, wandb,
torchmetrics, seaborn, and whatever else src/utils.py pulls in); the stand-ins
touch none of the arithmetic -- and stores inputs, parameters, outputs and
gradients as small .npz fixtures.  Fixture layout:  "P/<state_dict key>" parameters
and buffers, "in/<name>" inputs, "out/<name>" expected outputs, "grad/<key>"
expected gradients (of sum(out * cot) or of the loss), "cfg" a JSON string.

    python tools/gen_golden.py                       # rewrites every fixture
    python tools/gen_golden.py --only val_loop,auc   # only the named generators
"""
import importlib
import json
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class _Anything:
    """Inert stand-in: any attribute / call / subscript yields another stand-in."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __getitem__(self, k):
        return _Anything()

    def __iter__(self):
        return iter(())


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)

    def _missing(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Anything

    m.__getattr__ = _missing  # type: ignore[attr-defined]
    sys.modules[name] = m
    return m


def import_reference():
    class LightningModule(torch.nn.Module):
        def log(self, *a, **k):
            pass

    pl = _stub("pytorch_lightning", LightningModule=LightningModule, Callback=object, Trainer=_Anything)
    _stub("pytorch_lightning.callbacks")
    _stub("pytorch_lightning.loggers")
    pl.callbacks = sys.modules["pytorch_lightning.callbacks"]
    for name in ["ruamel", "ruamel.yaml", "wandb", "torchmetrics", "torchmetrics.classification", "seaborn",
                 "matplotlib", "matplotlib.pyplot", "matplotlib.colors", "matplotlib.ticker",
                 "matplotlib.patches", "matplotlib.lines", "h5py", "astropy", "astropy.io", "extinction",
                 "torchvision", "torchvision.transforms", "umap", "IPython"]:
        try:
            importlib.import_module(name)
        except Exception:
            _stub(name)
    sys.path.insert(0, REF)
    ref_loss = importlib.import_module("src.loss")
    ref_tr = importlib.import_module("src.transformer_utils")
    ref_mm = importlib.import_module("src.models_multimodal")
    return ref_loss, ref_tr, ref_mm


def save(name, cfg=None, **groups):
    arrays = {"cfg": np.array(json.dumps(cfg or {}))}
    for g, d in groups.items():
        for k, v in d.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            arrays[f"{g}/{k}"] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB")


def unit(n, d, g):
    x = torch.randn(n, d, generator=g)
    return x / x.norm(dim=-1, keepdim=True)


def sd_of(module):
    return {k: v.clone() for k, v in module.state_dict().items()}


def grads_of(module):
    return {k: p.grad.clone() for k, p in module.named_parameters() if p.grad is not None}


def randomise(module, g, scale=0.3):
    """Replace the default init by seeded values stored explicitly in the fixture (BN/LN gains
    near 1, running_var positive) so nothing depends on init order."""
    with torch.no_grad():
        for k, p in module.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * scale)
            if k.endswith("norm1.weight") or k.endswith("norm2.weight") or (
                    p.dim() == 1 and k.endswith(".weight") and "net." in k):
                p.add_(1.0)
        for k, b in module.named_buffers():
            if k.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif k.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)


def ragged_mask(b, t, g, nband=1):
    per = t // nband
    m = torch.zeros(b, t, dtype=torch.bool)
    for i in range(b):
        for k in range(nband):
            n = int(torch.randint(1, per + 1, (1,), generator=g))
            m[i, k * per:k * per + n] = True
    return m


# ----------------------------------------------------------------------------------------
def gen_loss(ref_loss):
    g = torch.Generator().manual_seed(11)
    eye = torch.eye(4)
    kat = {}
    kat["clip_I4_s0_b0"] = ref_loss.clip_loss(eye, eye, torch.tensor(0.0), torch.tensor(0.0))
    kat["clip_I4_ln10_bm10"] = ref_loss.clip_loss(eye, eye, torch.tensor(math.log(10.0)), torch.tensor(-10.0))
    kat["sigmoid_I4_ln10_bm10"] = ref_loss.sigmoid_loss(eye, eye, torch.tensor(math.log(10.0)),
                                                        torch.tensor(-10.0))
    save("loss_kat", out=kat)

    for n, d in [(8, 16), (32, 128), (256, 128)]:
        e1 = unit(n, d, g).requires_grad_()
        e2 = unit(n, d, g).requires_grad_()
        ls = torch.tensor(math.log(10.0), requires_grad=True)
        lb = torch.tensor(-10.0, requires_grad=True)
        loss = ref_loss.clip_loss(e1, e2, ls, lb)
        loss.backward()
        ins = {"e1": e1, "e2": e2, "logit_scale": ls, "logit_bias": lb}
        if n == 256:  # keep the fixture small: store fp16-exact inputs? no -- store as is (256 KB)
            pass
        save(f"loss_clip_n{n}", **{"in": ins, "out": {"loss": loss},
                                   "grad": {"e1": e1.grad, "e2": e2.grad, "logit_scale": ls.grad,
                                            "logit_bias": lb.grad}})
        e1s, e2s = e1.detach().clone().requires_grad_(), e2.detach().clone().requires_grad_()
        ls2 = torch.tensor(math.log(3.0), requires_grad=True)
        lb2 = torch.tensor(0.7, requires_grad=True)
        sl = ref_loss.sigmoid_loss(e1s, e2s, ls2, lb2)
        sl.backward()
        if n <= 32:
            save(f"loss_sigmoid_n{n}", **{"in": {"e1": e1s, "e2": e2s, "logit_scale": ls2, "logit_bias": lb2},
                                          "out": {"loss": sl},
                                          "grad": {"e1": e1s.grad, "e2": e2s.grad, "logit_scale": ls2.grad,
                                                   "logit_bias": lb2.grad}})

    # unequal lengths (n = min)
    e1, e2 = unit(6, 16, g), unit(9, 16, g)
    save("loss_clip_unequal", **{"in": {"e1": e1, "e2": e2, "logit_scale": torch.tensor(1.3),
                                        "logit_bias": torch.tensor(-0.5)},
                                 "out": {"loss": ref_loss.clip_loss(e1, e2, torch.tensor(1.3), torch.tensor(-0.5))}})

    # three-way multimodal, shared 0-dim scale/bias and per-pair vectors
    embs = [unit(8, 16, g).requires_grad_() for _ in range(3)]
    ls = torch.tensor(math.log(7.0), requires_grad=True)
    lb = torch.tensor(-10.0, requires_grad=True)
    loss = ref_loss.clip_loss_multimodal(embs, ls, lb)
    loss.backward()
    lsv = torch.tensor([0.5, 1.0, 1.5])
    lbv = torch.tensor([-1.0, 0.0, 2.0])
    save("loss_clip_multimodal3", **{
        "in": {"e0": embs[0], "e1": embs[1], "e2": embs[2], "logit_scale": ls, "logit_bias": lb,
               "scales_vec": lsv, "biases_vec": lbv},
        "out": {"loss": loss,
                "loss_vec": ref_loss.clip_loss_multimodal([e.detach() for e in embs], lsv, lbv),
                "sigmoid": ref_loss.sigmoid_loss_multimodal([e.detach() for e in embs], ls.detach(), lb.detach())},
        "grad": {"e0": embs[0].grad, "e1": embs[1].grad, "e2": embs[2].grad, "logit_scale": ls.grad,
                 "logit_bias": lb.grad}})


def gen_transformer(ref_tr):
    g = torch.Generator().manual_seed(23)
    b, t, e, h = 3, 12, 16, 4
    x = torch.randn(b, t, e, generator=g)
    cot = torch.randn(b, t, e, generator=g)
    masks = {"full": torch.ones(b, t, dtype=torch.bool), "ragged": ragged_mask(b, t, g)}
    masks["ragged"][2] = False  # a fully padded sample: softmax over -1e7 everywhere = uniform
    for mname, mask in masks.items():
        att = ref_tr.SelfAttention(e, heads=h)
        randomise(att, g)
        xi = x.clone().requires_grad_()
        y = att(xi, mask)
        (y * cot).sum().backward()
        save(f"attn_{mname}", cfg={"emb": e, "heads": h}, P=sd_of(att), **{
            "in": {"x": x, "mask": mask, "cot": cot}, "out": {"y": y},
            "grad": {"x": xi.grad, **grads_of(att)}})

        blk = ref_tr.TransformerBlock(e, h, ff_hidden_mult=4, dropout=0.0)
        randomise(blk, g)
        xi = x.clone().requires_grad_()
        y = blk(xi, mask)
        (y * cot).sum().backward()
        save(f"block_{mname}", cfg={"emb": e, "heads": h}, P=sd_of(blk), **{
            "in": {"x": x, "mask": mask, "cot": cot}, "out": {"y": y},
            "grad": {"x": xi.grad, **grads_of(blk)}})

    tt = torch.rand(b, t, generator=g) * 100.0
    pe = ref_tr.TimePositionalEncoding(e, 20583.37)(tt)
    save("timeenc", cfg={"emb": e, "norm": 20583.37}, **{"in": {"t": tt}, "out": {"pe": pe}})

    n_out = 8
    for agg in ["mean", "max", "attn", "pretraining"]:
        for nband in [1, 2]:
            for mname in ["full", "ragged"]:
                mask = torch.ones(b, t, dtype=torch.bool) if mname == "full" else ragged_mask(b, t, g, nband)
                kw = dict(emb=e, heads=h, depth=2, dropout=0.0)
                m = ref_tr.TransformerWithTimeEmbeddings(n_out=n_out, nband=nband, agg=agg,
                                                         time_norm=20583.37, **kw)
                randomise(m, g)
                xv = torch.randn(b, t, 1, generator=g)
                tv = torch.sort(torch.rand(b, t, generator=g) * 100.0, dim=1)[0]
                y = m(xv, tv, mask)
                c = torch.randn(y.shape, generator=g)
                (y * c).sum().backward()
                save(f"tenc_{agg}_nb{nband}_{mname}",
                     cfg={"emb": e, "heads": h, "depth": 2, "time_norm": 20583.37, "nband": nband,
                          "agg": agg, "n_out": n_out},
                     P=sd_of(m), **{"in": {"x": xv, "t": tv, "mask": mask, "cot": c}, "out": {"y": y},
                                    "grad": grads_of(m)})


def gen_transformer_e32(ref_tr):
    """The reference's spectrum-transformer width (emb 32, 2 heads: transformer_kwargs_spectral of configs/maven-lite.yaml): the
    shapes whose feed-forward half runs as the fused kernels of csrc/ffn_planes.hip and -- beyond 128 tokens -- whose attention
    runs on the bf16 planes.  Same generators as gen_transformer, their own seed (the fixtures of gen_transformer stay byte-identical)."""
    g = torch.Generator().manual_seed(61)
    e, h = 32, 2
    for mname, (b, t) in {"full": (3, 12), "ragged": (3, 12), "long": (2, 160)}.items():
        x = torch.randn(b, t, e, generator=g)
        cot = torch.randn(b, t, e, generator=g)
        mask = torch.ones(b, t, dtype=torch.bool) if mname == "full" else ragged_mask(b, t, g)
        if mname == "ragged":
            mask[2] = False
        blk = ref_tr.TransformerBlock(e, h, ff_hidden_mult=4, dropout=0.0)
        randomise(blk, g)
        xi = x.clone().requires_grad_()
        y = blk(xi, mask)
        (y * cot).sum().backward()
        save(f"block_e32_{mname}", cfg={"emb": e, "heads": h}, P=sd_of(blk), **{
            "in": {"x": x, "mask": mask, "cot": cot}, "out": {"y": y},
            "grad": {"x": xi.grad, **grads_of(blk)}})
    n_out = 8
    for mname, (b, t, nband, agg) in {"mean_nb1_e32_long": (2, 160, 1, "mean"), "attn_nb2_e32_ragged": (3, 24, 2, "attn")}.items():
        mask = ragged_mask(b, t, g, nband)
        m = ref_tr.TransformerWithTimeEmbeddings(n_out=n_out, nband=nband, agg=agg, time_norm=17945.14, emb=e, heads=h, depth=3,
                                                 dropout=0.0)
        randomise(m, g)
        xv = torch.randn(b, t, 1, generator=g)
        tv = torch.sort(torch.rand(b, t, generator=g) * 6000.0 + 3000.0, dim=1)[0]
        y = m(xv, tv, mask)
        c = torch.randn(y.shape, generator=g)
        (y * c).sum().backward()
        save(f"tenc_{mname}", cfg={"emb": e, "heads": h, "depth": 3, "time_norm": 17945.14, "nband": nband, "agg": agg, "n_out": n_out},
             P=sd_of(m), **{"in": {"x": xv, "t": tv, "mask": mask, "cot": c}, "out": {"y": y}, "grad": grads_of(m)})


def gen_convmixer_mlp(ref_mm):
    g = torch.Generator().manual_seed(37)
    for name, (hw, p) in {"p4": (16, 4), "p10floor": (23, 10)}.items():
        kw = dict(dim=8, depth=2, channels=3, kernel_size=5, patch_size=p, n_out=8, dropout_prob=0.0)
        for mode in ["train", "eval"]:
            m = ref_mm.ConvMixer(**kw)
            randomise(m, g)
            m.train(mode == "train")
            before = sd_of(m)
            x = torch.rand(5, 3, hw, hw, generator=g)
            xi = x.clone().requires_grad_()
            y = m(xi)
            c = torch.randn(y.shape, generator=g)
            (y * c).sum().backward()
            after = {k: v for k, v in sd_of(m).items() if "running" in k or "num_batches" in k}
            save(f"convmixer_{name}_{mode}", cfg={**kw, "mode": mode}, P=before,
                 **{"in": {"x": x, "cot": c}, "out": {"y": y}, "grad": {"x": xi.grad, **grads_of(m)},
                    "stats": after})
    m = ref_mm.MLP(input_dim=12, hidden_dim=16, output_dim=8, num_layers=2, dropout=0.0)
    randomise(m, g)
    x = torch.randn(6, 12, generator=g)
    xi = x.clone().requires_grad_()
    y = m(xi)
    c = torch.randn(y.shape, generator=g)
    (y * c).sum().backward()
    save("mlp", cfg={"num_layers": 2}, P=sd_of(m),
         **{"in": {"x": x, "cot": c}, "out": {"y": y}, "grad": {"x": xi.grad, **grads_of(m)}})


def _batch(g, b, combos, hw=16, t_lc=12, t_sp=10, nband=2, n_classes=5):
    x_img = torch.rand(b, 3, hw, hw, generator=g) if "host_galaxy" in combos else None
    x_lc = torch.randn(b, t_lc, generator=g)
    per = t_lc // nband
    t_lcv = torch.cat([torch.sort(torch.rand(b, per, generator=g) * 100.0, dim=1)[0] for _ in range(nband)], 1)
    m_lc = ragged_mask(b, t_lc, g, nband)
    x_sp = torch.randn(b, t_sp, generator=g)
    t_spv = torch.sort(torch.rand(b, t_sp, generator=g) * 6000.0 + 3000.0, dim=1)[0]
    m_sp = ragged_mask(b, t_sp, g)
    red = torch.rand(b, generator=g)
    cls = torch.randint(0, n_classes, (b,), generator=g)
    return (x_img, x_lc, t_lcv, m_lc, x_sp, t_spv, m_sp, red, cls)


def gen_clip(ref_mm):
    g = torch.Generator().manual_seed(53)
    tk = dict(n_out=8, emb=16, heads=4, depth=2, dropout=0.0, time_norm=20583.37, agg="mean")
    sk = dict(n_out=8, emb=8, heads=2, depth=3, dropout=0.0, time_norm=17945.14, agg="mean")
    ck = dict(dim=8, depth=2, channels=3, kernel_size=5, patch_size=4, n_out=8, dropout_prob=0.0)
    mk = dict(input_dim=8, hidden_dim=16, num_layers=2, dropout=0.0)
    cases = {
        "clip_img_lc": ["host_galaxy", "lightcurve"],
        "clip_lc_sp": ["lightcurve", "spectral"],
        "clip_3tower": ["spectral", "host_galaxy", "lightcurve"],   # order given != order used
        "clip_4tower": ["host_galaxy", "lightcurve", "spectral", "meta"],
    }
    for name, combos in cases.items():
        for loss in (["softmax", "sigmoid"] if name == "clip_lc_sp" else ["softmax"]):
            model = ref_mm.LightCurveImageCLIP(
                enc_dim=16, logit_scale=10.0, nband=2, transformer_kwargs=tk, transformer_spectral_kwargs=sk,
                conv_kwargs=ck, meta_kwargs=mk, combinations=combos, optimizer_kwargs={"weight_decay": 1e-3},
                lr=1e-2, loss=loss)
            randomise(model, g)
            with torch.no_grad():
                model.logit_scale.fill_(math.log(10.0))
                model.logit_bias.fill_(-10.0)
            model.train()
            before = sd_of(model)
            b = 6
            batch = _batch(g, b, combos)
            embs = model(*batch)
            model.zero_grad()
            loss_v = model.training_step(batch, 0)
            loss_v.backward()
            cfg = {"combinations": combos, "nband": 2, "transformer_kwargs": tk,
                   "transformer_spectral_kwargs": sk, "conv_kwargs": ck, "meta_kwargs": mk, "enc_dim": 16,
                   "loss": loss, "lr": 1e-2, "weight_decay": 1e-3}
            ins = {k: v for k, v in zip(["x_img", "x_lc", "t_lc", "mask_lc", "x_sp", "t_sp", "mask_sp",
                                         "redshift", "classification"], batch) if v is not None}
            outs = {f"emb{i}": e for i, e in enumerate(embs)}
            outs["loss"] = loss_v
            suffix = "" if loss == "softmax" else "_sigmoid"
            save(name + suffix, cfg=cfg, P=before, **{"in": ins, "out": outs, "grad": grads_of(model)})

    # harness row H: 3 optimiser steps with RAdam as configure_optimizers builds it, two batches cycled
    combos = ["host_galaxy", "lightcurve"]
    model = ref_mm.LightCurveImageCLIP(
        enc_dim=16, logit_scale=10.0, nband=2, transformer_kwargs=tk, transformer_spectral_kwargs=sk,
        conv_kwargs=ck, meta_kwargs=mk, combinations=combos, optimizer_kwargs={"weight_decay": 1e-3},
        lr=1e-2, loss="softmax")
    randomise(model, g)
    with torch.no_grad():
        model.logit_scale.fill_(math.log(10.0))
        model.logit_bias.fill_(-10.0)
    model.train()
    before = sd_of(model)
    opt = model.configure_optimizers()["optimizer"]
    batches = [_batch(g, 6, combos) for _ in range(2)]
    losses = []
    n_steps = 8  # crosses RAdam's rho_t > 5 switch (step 6 with beta2 = 0.999)
    for step in range(n_steps):
        opt.zero_grad()
        lv = model.training_step(batches[step % 2], step)
        lv.backward()
        opt.step()
        losses.append(lv.detach())
    ins = {}
    for bi, batch in enumerate(batches):
        for k, v in zip(["x_img", "x_lc", "t_lc", "mask_lc", "x_sp", "t_sp", "mask_sp", "redshift",
                         "classification"], batch):
            if v is not None:
                ins[f"b{bi}.{k}"] = v
    cfg = {"combinations": combos, "nband": 2, "transformer_kwargs": tk, "transformer_spectral_kwargs": sk,
           "conv_kwargs": ck, "meta_kwargs": mk, "enc_dim": 16, "loss": "softmax", "lr": 1e-2,
           "weight_decay": 1e-3, "n_steps": n_steps}
    save("harness_radam", cfg=cfg, P=before, **{"in": ins, "out": {"losses": torch.stack(losses)},
                                               "after": sd_of(model)})


def gen_real_checkpoint(ref_mm):
    """a SHIPPED reference checkpoint (real trained weights, Lightning 2.2.3 format)
    loaded strictly into the reference module; embeddings + loss on a fixed synthetic lc + spectrum batch."""
    ckpt = os.path.join(REF, "models", "clip_finetune", "absurd-sweep-1", "epoch=118-step=12495.ckpt")
    sd = torch.load(ckpt, map_location="cpu", weights_only=True)["state_dict"]
    tk = dict(n_out=32, emb=64, heads=8, depth=5, dropout=0.0, time_norm=20583.369161312577, agg="mean")
    sk = dict(n_out=32, emb=32, heads=2, depth=13, dropout=0.0, time_norm=17945.142213594805, agg="mean")
    model = ref_mm.LightCurveImageCLIP(enc_dim=128, logit_scale=19.545966923442453, nband=2, transformer_kwargs=tk,
                                       transformer_spectral_kwargs=sk, combinations=["lightcurve", "spectral"],
                                       loss="softmax")
    missing = model.load_state_dict(sd, strict=True)
    print("real checkpoint:", missing, f"exp(logit_scale) = {float(model.logit_scale.exp()):.3f}")
    model.eval()
    g = torch.Generator().manual_seed(71)
    b = 8
    batch = _batch(g, b, ["lightcurve", "spectral"], t_lc=200, t_sp=220, nband=2)
    with torch.no_grad():
        embs = model(*batch)
        loss = model.training_step(batch, 0)
    cfg = {"combinations": ["lightcurve", "spectral"], "nband": 2, "transformer_kwargs": tk,
           "transformer_spectral_kwargs": sk, "conv_kwargs": None, "meta_kwargs": None, "enc_dim": 128,
           "loss": "softmax", "lr": 1e-4, "weight_decay": 0.0, "checkpoint": "models/clip_finetune/absurd-sweep-1/"
           "epoch=118-step=12495.ckpt"}
    ins = {k: v for k, v in zip(["x_img", "x_lc", "t_lc", "mask_lc", "x_sp", "t_sp", "mask_sp", "redshift",
                                 "classification"], batch) if v is not None}
    save("real_ckpt_lc_sp", cfg=cfg, P={k: v for k, v in sd.items()},
         **{"in": ins, "out": {"emb0": embs[0], "emb1": embs[1], "loss": loss}})


def gen_auc():
    """Row f2: the reference's retrieval metric (src/utils.py:380-426) on correlated random embeddings."""
    ref_utils = importlib.import_module("src.utils")
    g = torch.Generator().manual_seed(91)
    out, ins = {}, {}
    for n, d, noise in [(50, 16, 0.6), (137, 32, 1.5)]:
        e1 = torch.randn(n, d, generator=g)
        e2 = e1 + noise * torch.randn(n, d, generator=g)
        thr, frac = ref_utils.get_ROC_data(e1, e2)
        ins[f"e1_{n}"], ins[f"e2_{n}"] = e1, e2
        out[f"thresholds_{n}"], out[f"fraction_{n}"] = thr, frac
        out[f"auc_{n}"] = np.float64(ref_utils.get_AUC(e1, e2))
    save("auc", **{"in": ins, "out": out})


def gen_pretraining():
    """Row f4: MaskedLightCurveEncoder forward + masked MSE + gradients with explicit masks from the reference's
    own get_continous_random_mask (python RNG seeded)."""
    import random
    ref_pt = importlib.import_module("src.models_pretraining")
    g = torch.Generator().manual_seed(101)
    random.seed(5)
    b, t, nband = 4, 20, 2
    tk = dict(n_out=1, emb=16, heads=4, depth=2, dropout=0.0, time_norm=20583.37)
    m = ref_pt.MaskedLightCurveEncoder(f_mask=0.3, nband=nband, transformer_kwargs=tk, lr_scheduler_kwargs={"step_size": 10})
    randomise(m, g)
    x = torch.randn(b, t, generator=g)
    tt = torch.cat([torch.sort(torch.rand(b, t // nband, generator=g) * 100, dim=1)[0] for _ in range(nband)], 1)
    pad = ragged_mask(b, t, g, nband)
    for i in range(b):                      # at least 4 observed points per band so something gets hidden
        for k in range(nband):
            pad[i, k * (t // nband): k * (t // nband) + 4] = True
    mask_in, mask_pred = ref_pt.get_continous_random_mask(pad, nband, f_mask=0.3)
    xm = x.clone()
    xm[~mask_in] = 0
    pred = m(xm, tt, mask=pad)
    loss = torch.nn.MSELoss()(x[mask_pred], pred[mask_pred])
    loss.backward()
    save("pretraining", cfg={"nband": nband, "f_mask": 0.3, "transformer_kwargs": tk}, P=sd_of(m),
         **{"in": {"x": x, "t": tt, "padding_mask": pad, "mask_in": mask_in, "mask_pred": mask_pred},
            "out": {"pred": pred, "loss": loss}, "grad": grads_of(m)})


def gen_random_masks():
    """Row f4, mask helpers: the reference's get_random_mask (torch RNG seeded) and get_continous_random_mask (python RNG
    seeded) on ragged padding masks, incl. 3 bands and a sample with a single observed point per band."""
    import random
    ref_pt = importlib.import_module("src.models_pretraining")
    g = torch.Generator().manual_seed(171)
    out = {}
    for tag, (b, t, nband, f) in {"a": (6, 24, 2, 0.3), "b": (5, 30, 3, 0.15), "c": (4, 20, 1, 0.5)}.items():
        pad = ragged_mask(b, t, g, nband)
        pad[0] = False
        for k in range(nband):
            pad[0, k * (t // nband)] = True          # a single observed point per band: nothing to hide
        torch.manual_seed(1000 + b)
        m, mp = ref_pt.get_random_mask(pad, f_mask=f)
        random.seed(2000 + b)
        cm, cmp_ = ref_pt.get_continous_random_mask(pad, nband, f_mask=f)
        out[tag] = {"padding_mask": pad, "nband": torch.tensor(nband), "f_mask": torch.tensor(f), "torch_seed": torch.tensor(1000 + b),
                    "python_seed": torch.tensor(2000 + b), "mask": m, "mask_pred": mp, "cmask": cm, "cmask_pred": cmp_}
    save("random_masks", **out)


def gen_val_loop(ref_mm):
    """Row a15: the reference's validation hooks end to end (src/models_multimodal.py:415-556) --
    on_validation_start -> validation_step x 3 (the last batch short) -> on_validation_epoch_end -- with every
    `self.log(name, value)` call captured: per-batch val_loss and AUC_val (two modalities) / AUC_val1..3 (three)."""
    g = torch.Generator().manual_seed(131)
    tk = dict(n_out=8, emb=16, heads=4, depth=2, dropout=0.0, time_norm=20583.37, agg="mean")
    sk = dict(n_out=8, emb=8, heads=2, depth=3, dropout=0.0, time_norm=17945.14, agg="mean")
    ck = dict(dim=8, depth=2, channels=3, kernel_size=5, patch_size=4, n_out=8, dropout_prob=0.0)
    mk = dict(input_dim=8, hidden_dim=16, num_layers=2, dropout=0.0)
    for name, combos in {"val_loop_lc_sp": ["lightcurve", "spectral"],
                         "val_loop_3tower": ["host_galaxy", "lightcurve", "spectral"]}.items():
        model = ref_mm.LightCurveImageCLIP(
            enc_dim=16, logit_scale=10.0, nband=2, transformer_kwargs=tk, transformer_spectral_kwargs=sk,
            conv_kwargs=ck, meta_kwargs=mk, combinations=combos, optimizer_kwargs={"weight_decay": 1e-3},
            lr=1e-2, loss="softmax")
        randomise(model, g)
        with torch.no_grad():
            model.logit_scale.fill_(math.log(10.0))
            model.logit_bias.fill_(-10.0)
        logged = []
        model.log = lambda key, value, **kw: logged.append((key, float(value)))
        model.eval()
        batches = [_batch(g, b, combos) for b in (6, 6, 4)]
        with torch.no_grad():
            model.on_validation_start()
            for i, batch in enumerate(batches):
                model.validation_step(batch, i)
            model.on_validation_epoch_end()
        assert model.embs_list is None
        ins = {}
        for bi, batch in enumerate(batches):
            for k, v in zip(["x_img", "x_lc", "t_lc", "mask_lc", "x_sp", "t_sp", "mask_sp", "redshift",
                             "classification"], batch):
                if v is not None:
                    ins[f"b{bi}.{k}"] = v
        out = {"val_losses": torch.tensor([v for k, v in logged if k == "val_loss"], dtype=torch.float64)}
        for k, v in logged:
            if k.startswith("AUC_val"):
                out[k] = np.float64(v)
        cfg = {"combinations": combos, "nband": 2, "transformer_kwargs": tk, "transformer_spectral_kwargs": sk,
               "conv_kwargs": ck, "meta_kwargs": mk, "enc_dim": 16, "loss": "softmax", "lr": 1e-2,
               "weight_decay": 1e-3, "batch_sizes": [6, 6, 4], "logged_keys": [k for k, _ in logged]}
        save(name, cfg=cfg, P=sd_of(model), **{"in": ins, "out": out})


def gen_augment():
    """Row f3: NoisyDataLoader.__iter__ (ref src/dataloader.py:88-240) run for the combinations whose branches need no
    torchvision -- {lightcurve} (:119-126), {spectral} (:128-137), {spectral, lightcurve} (:215-238) -- under
    torch.manual_seed.  The loader is given its own torch.Generator for the sampler / worker seed, so the GLOBAL stream the
    branch draws its Gaussian fields from starts right at the seed: the fields are replayed here, asserted to reproduce the
    loader's output bit for bit, and stored with it.  The image branches (:96-114: uniform noise + torchvision's
    RandomRotation) cannot be run in this image (torchvision is absent) and stay unpinned."""
    import importlib
    for name in ["astropy.cosmology", "PIL", "PIL.Image"]:       # inert stand-ins: none of them is touched by __iter__
        try:
            importlib.import_module(name)
        except Exception:
            _stub(name)
    dl = importlib.import_module("src.dataloader")
    from torch.utils.data import TensorDataset
    g = torch.Generator().manual_seed(77)
    n, t_lc, t_sp = 10, 24, 40

    def series(t):
        x = torch.randn(n, t, generator=g)
        tt = torch.sort(torch.rand(n, t, generator=g) * 100.0, dim=1)[0]
        mask = torch.rand(n, t, generator=g) > 0.2
        err = torch.rand(n, t, generator=g) * 0.5 + 0.01
        return x, tt, mask, err

    mag, time, mask, magerr = series(t_lc)
    spec, freq, maskspec, specerr = series(t_sp)
    red, cls = torch.rand(n, generator=g), torch.randint(0, 5, (n,), generator=g)
    cases = {"lightcurve": ((mag, time, mask, magerr, red, cls), ["lightcurve"], 0.7, 4),
             "spectral": ((spec, freq, maskspec, specerr, red, cls), ["spectral"], 1.3, 5),
             "lc_sp": ((mag, time, mask, magerr, spec, freq, maskspec, specerr, red, cls), ["spectral", "lightcurve"], 0.9, 4)}
    for name, (tensors, combos, level, bs) in cases.items():
        loader = dl.NoisyDataLoader(TensorDataset(*tensors), batch_size=bs, noise_level_img=0.25, noise_level_mag=level,
                                    combinations=list(combos), shuffle=False, generator=torch.Generator().manual_seed(5))
        seed = 1000 + len(name)
        torch.manual_seed(seed)
        outs = list(loader)
        # replay of the global stream: per batch, one field per noisy series in the order the branch draws them
        torch.manual_seed(seed)
        ins, out = {}, {}
        for bi, o in enumerate(outs):
            lo, hi = bi * bs, min(n, (bi + 1) * bs)
            if "lightcurve" in combos:
                f = torch.randn_like(mag[lo:hi])
                assert torch.equal(o[1], mag[lo:hi] + f * magerr[lo:hi] * level), name
                assert torch.equal(o[2], time[lo:hi]) and torch.equal(o[3], mask[lo:hi])
                ins[f"b{bi}.x_lc"], ins[f"b{bi}.err_lc"], ins[f"b{bi}.field_lc"], out[f"b{bi}.x_lc"] = mag[lo:hi], magerr[lo:hi], f, o[1]
            if "spectral" in combos:
                f = torch.randn_like(spec[lo:hi])
                assert torch.equal(o[4], spec[lo:hi] + f * specerr[lo:hi] * level), name
                ins[f"b{bi}.x_sp"], ins[f"b{bi}.err_sp"], ins[f"b{bi}.field_sp"], out[f"b{bi}.x_sp"] = spec[lo:hi], specerr[lo:hi], f, o[4]
            assert o[0] is None and len(o) == 9
        save(f"augment_{name}", cfg={"combinations": combos, "noise_level_mag": level, "batch_size": bs, "batches": len(outs),
                                     "seed": seed}, **{"in": ins, "out": out})


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(4)
    ref_loss, ref_tr, ref_mm = import_reference()
    only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
    jobs = {"loss": lambda: gen_loss(ref_loss), "transformer": lambda: gen_transformer(ref_tr), "transformer_e32": lambda: gen_transformer_e32(ref_tr),
            "convmixer_mlp": lambda: gen_convmixer_mlp(ref_mm), "clip": lambda: gen_clip(ref_mm),
            "real_checkpoint": lambda: gen_real_checkpoint(ref_mm), "auc": gen_auc, "pretraining": gen_pretraining, "random_masks": gen_random_masks,
            "val_loop": lambda: gen_val_loop(ref_mm), "augment": gen_augment}
    for name, job in jobs.items():        # each generator seeds its own torch.Generator: independent of the others
        if only is None or name in only:
            job()


if __name__ == "__main__":
    main()

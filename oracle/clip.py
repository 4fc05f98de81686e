"""Oracle: the CLIP module's forward / training step / optimiser (TEST INFRASTRUCTURE).

Functional restatement of LightCurveImageCLIP's contrastive branch
(/root/reference/src/models_multimodal.py:259-304, :312-366, :306-310) over a flat
state_dict-keyed mapping; pinned by tests/golden/clip_*.npz and radam_*.npz.
"""
import math

import torch

from . import encoders as enc
from . import loss as L


def l2_normalise(x):
    """x / ||x||_2 with no epsilon -- ref models_multimodal.py:279, :286, :293, :304."""
    return x / torch.sqrt((x * x).sum(dim=-1, keepdim=True))


def embeddings(P, cfg, batch, training=True, stats_out=None):
    """List of (B, enc_dim) unit vectors in the FIXED order host_galaxy, lightcurve,
    spectral, meta (ref :259-273), whatever order `combinations` was given in.

    cfg keys: combinations, nband, transformer_kwargs, transformer_spectral_kwargs,
    conv_kwargs, meta_kwargs (same dicts the reference constructor takes).
    batch: the reference's 9-tuple (x_img, x_lc, t_lc, mask_lc, x_sp, t_sp, mask_sp, redshift, cls).
    """
    x_img, x_lc, t_lc, mask_lc, x_sp, t_sp, mask_sp, redshift, cls = batch
    combos = set(cfg["combinations"])
    out = []
    if "host_galaxy" in combos:
        ck = cfg["conv_kwargs"]
        h = enc.convmixer(P, "image_encoder.", x_img, depth=ck["depth"], patch_size=ck["patch_size"],
                          training=training, stats_out=stats_out)
        out.append(l2_normalise(enc.linear(P, "image_projection", h)))
    if "lightcurve" in combos:
        tk = cfg["transformer_kwargs"]
        h = enc.transformer_with_time_embeddings(
            P, "lightcurve_encoder.", x_lc[..., None], t_lc, mask_lc, emb=tk["emb"], heads=tk["heads"],
            depth=tk["depth"], time_norm=tk["time_norm"], nband=cfg.get("nband", 1),
            agg=tk.get("agg", "mean"))
        out.append(l2_normalise(enc.linear(P, "lightcurve_projection", h)))
    if "spectral" in combos:
        tk = cfg["transformer_spectral_kwargs"]
        h = enc.transformer_with_time_embeddings(
            P, "spectral_encoder.", x_sp[..., None], t_sp, mask_sp, emb=tk["emb"], heads=tk["heads"],
            depth=tk["depth"], time_norm=tk["time_norm"], nband=1, agg=tk.get("agg", "mean"))
        out.append(l2_normalise(enc.linear(P, "spectral_projection", h)))
    if "meta" in combos:
        mk = cfg["meta_kwargs"]
        half = mk["input_dim"] // 2
        h = torch.cat([P["class_emb.weight"][cls], redshift[:, None].repeat(1, half)], dim=-1)
        out.append(l2_normalise(enc.mlp(P, "meta_encoder.", h, mk["num_layers"])))
    return out


def training_loss(P, cfg, batch, loss="softmax", training=True, stats_out=None):
    """ref training_step :353-361: pairwise-summed loss on the forward's embeddings."""
    embs = embeddings(P, cfg, batch, training=training, stats_out=stats_out)
    fn = L.clip_loss_multimodal if loss == "softmax" else L.sigmoid_loss_multimodal
    return fn(embs, P["logit_scale"], P["logit_bias"])


class RAdam:
    """torch.optim.RAdam restated (the optimiser the reference builds at :306-310 with
    torch defaults: betas (0.9, 0.999), eps 1e-8, L2 weight decay folded into the gradient).
    Checked against torch.optim.RAdam in tests/test_oracle_golden.py."""

    def __init__(self, params, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.params = list(params)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        b1, b2 = self.betas
        t = self.t
        c1 = 1.0 - b1 ** t
        c2 = 1.0 - b2 ** t
        rho_inf = 2.0 / (1.0 - b2) - 1.0
        rho_t = rho_inf - 2.0 * t * (b2 ** t) / c2
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad + self.wd * p if self.wd != 0.0 else p.grad
            m.mul_(b1).add_(g, alpha=1.0 - b1)
            v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
            m_hat = m / c1
            if rho_t > 5.0:
                rect = math.sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf
                                 / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t))
                p.add_(m_hat * (math.sqrt(c2) / (v.sqrt() + self.eps)), alpha=-self.lr * rect)
            else:
                p.add_(m_hat, alpha=-self.lr)

    def zero_grad(self):
        for p in self.params:
            p.grad = None


def roc_data(embs1, embs2):
    """ref src/utils.py:380-411 restated without the per-row loop: rank of the partner = number of rows of
    embs1 strictly more similar (cosine) to embs2[i] than embs1[i]; 100 thresholds, top-int(thr * N) hits."""
    import numpy as np
    a = embs1 / embs1.norm(dim=-1, keepdim=True)
    b = embs2 / embs2.norm(dim=-1, keepdim=True)
    sim = b @ a.T                                           # row i: similarities of embs2[i] to every embs1[j]
    ranks = (sim > sim.diagonal()[:, None]).sum(dim=1).numpy()
    n = len(ranks)
    thresholds = np.linspace(0, 1, 100)
    top = np.array([int(t * n) for t in thresholds])
    return thresholds, (ranks[None, :] < top[:, None]).sum(axis=1) / n, ranks


def auc(embs1, embs2):
    import numpy as np
    t, f, _ = roc_data(embs1, embs2)
    return np.trapezoid(f, t) if hasattr(np, "trapezoid") else np.trapz(f, t)

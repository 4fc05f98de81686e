"""Masked-light-curve pretraining with the reference's surface, src/models_pretraining.py:
`get_random_mask`, `get_continous_random_mask` (masks as set operations on the device; the random draws are the
reference's -- one per sample / band, in its order -- so a seeded run hides the same points) and
`MaskedLightCurveEncoder` = TransformerWithTimeEmbeddings(agg="pretraining") + Linear(emb, 1), trained with an
MSE on the hidden points.  The transformer, the read-out GEMM and the masked MSE run on libmsn_hip."""
import random
from typing import Dict, Tuple

import torch
import torch.nn as nn

from . import functional as F_
from . import ops
from ._lib import check, lib, ptr, stream_ptr
from .transformer_utils import TransformerWithTimeEmbeddings


def _hidden_counts(observed_counts, f_mask):
    """floor(n_observed * f_mask) in double precision, as Python's int(n * f) gives it."""
    return (observed_counts.to(torch.float64) * float(f_mask)).floor().to(torch.int64)


def get_random_mask(padding_mask, f_mask=0.15):
    """Hide a random fraction of the observed points of every sample (ref src/models_pretraining.py:17-55).
    Returns (mask, mask_pred): `mask` = the padding mask without the hidden points, `mask_pred` = the hidden points.

    Set formulation: draw, per sample, one ranking of its observed points (ONE torch.randperm(n_observed) call per sample,
    in sample order -- the reference's draws, so a seeded run hides the same points), call the points ranked below
    n_hide = floor(f_mask * n_observed) "hidden", and build both outputs from the hidden set with two mask operations on
    the device the padding mask lives on."""
    pad = padding_mask.to(torch.bool)
    B, T = pad.shape
    n_obs = pad.sum(dim=1)
    n_hide = _hidden_counts(n_obs, f_mask)
    order = torch.full((B, T), T, dtype=torch.int64)              # rank of every observed point in its sample's draw
    pad_host = pad.cpu()
    for i, n in enumerate(n_obs.tolist()):
        ranks = torch.empty(n, dtype=torch.int64)
        ranks[torch.randperm(n)] = torch.arange(n)                 # the j-th observed point has rank ranks[j]
        order[i, pad_host[i]] = ranks
    hidden = order.to(pad.device) < n_hide.to(pad.device)[:, None]
    return pad & ~hidden, pad & hidden


def get_continous_random_mask(padding_mask, nbands, f_mask=0.15):
    """Hide one random CONTIGUOUS run of observed points per band (ref src/models_pretraining.py:58-98): in band k of
    sample i, with n observed points (packed at the start of the band) and h = floor(f_mask * n), the run starts at a
    uniformly drawn offset in [0, n - h] (ONE random.randint per (sample, band), sample-major: the reference's draws) and
    is h long.  Both outputs are then interval tests against the drawn starts, on the padding mask's device."""
    pad = padding_mask.to(torch.bool)
    B, T = pad.shape
    band = T // nbands
    n_obs = pad[:, :band * nbands].reshape(B, nbands, band).sum(dim=2)
    n_hide = _hidden_counts(n_obs, f_mask)
    starts = torch.tensor([[random.randint(band * k, band * k + n - h) for k, (n, h) in enumerate(zip(ns, hs))]
                           for ns, hs in zip(n_obs.tolist(), n_hide.tolist())], dtype=torch.int64).reshape(B, nbands)
    pos = torch.arange(T, device=pad.device)[None, :].expand(B, T)
    which = torch.clamp(pos // band, max=nbands - 1)               # band of every position (a ragged tail joins the last band)
    lo = torch.gather(starts.to(pad.device), 1, which)
    hi = lo + torch.gather(n_hide.to(pad.device), 1, which)
    inside = (pos >= lo) & (pos < hi) & (pos < band * nbands)
    mask_pred = pad & inside
    mask_pred[:, band * nbands:] = pad[:, band * nbands:]          # positions beyond the last whole band are left as they are
    return pad & ~inside, mask_pred


class _MaskedMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, select_u8):
        pred, target = pred.contiguous(), target.contiguous().float()
        stats = torch.empty(2, dtype=torch.float32, device=pred.device)
        check(lib().msn_masked_mse_fwd(ptr(ops._f32c(pred, "pred")), ptr(target), ptr(select_u8), pred.numel(), ptr(stats),
                                       stream_ptr()), "msn_masked_mse_fwd")
        ctx.sel = select_u8
        ctx.save_for_backward(pred, target, stats)
        return stats[0]

    @staticmethod
    def backward(ctx, g):
        pred, target, stats = ctx.saved_tensors
        d = torch.empty_like(pred)
        g = g.to(torch.float32).reshape(()).contiguous()
        check(lib().msn_masked_mse_bwd(ptr(pred), ptr(target), ptr(ctx.sel), pred.numel(), ptr(stats), ptr(g), ptr(d),
                                       stream_ptr()), "msn_masked_mse_bwd")
        return d, None, None


def masked_mse(pred, target, select):
    """mean((pred - target)^2) over the elements where `select` is true == nn.MSELoss()(pred[select], target[select])."""
    return _MaskedMSE.apply(pred, target, ops._mask_u8(select))


class MaskedLightCurveEncoder(nn.Module):
    """ref src/models_pretraining.py:101-259 (Lightning hooks as plain methods)."""

    def __init__(self, f_mask: float = 0.2, nband: int = 1, transformer_kwargs: Dict = None, optimizer_kwargs: Dict = None,
                 lr_scheduler_kwargs: Dict = None, lr: float = 1e-3):
        super().__init__()
        transformer_kwargs = dict(transformer_kwargs or {"n_out": 1, "emb": 128, "heads": 2, "depth": 4})
        self.nband, self.lr, self.f_mask = nband, lr, f_mask
        self.optimizer_kwargs = dict(optimizer_kwargs or {})
        self.lr_scheduler_kwargs = dict(lr_scheduler_kwargs or {})
        self.net = TransformerWithTimeEmbeddings(nband=nband, agg="pretraining", **transformer_kwargs)
        self.last_layer = nn.Linear(transformer_kwargs["emb"], 1)
        self.logged = {}

    def log(self, name, value, **kwargs):
        # detached: a logged loss must not keep its autograd graph alive into the next step
        self.logged[name] = value.detach() if torch.is_tensor(value) else value

    def forward(self, x, t, mask=None):
        h = self.net(x[..., None], t, mask)                                    # (B, T, emb), padded tokens zeroed
        return F_.linear(h, self.last_layer.weight, self.last_layer.bias).squeeze(2)

    def configure_optimizers(self):
        """RAdam + StepLR stepped once per epoch (ref src/models_pretraining.py:167-189); trainer.Trainer steps the
        scheduler it finds under "lr_scheduler" after every training epoch, as Lightning does."""
        from .optim import RAdam
        optimizer = RAdam(self.parameters(), lr=self.lr, **self.optimizer_kwargs)
        out = {"optimizer": optimizer}
        if self.lr_scheduler_kwargs:
            out["lr_scheduler"] = {"scheduler": torch.optim.lr_scheduler.StepLR(optimizer, **self.lr_scheduler_kwargs),
                                   "monitor": "val_loss", "interval": "epoch", "frequency": 1}
        return out

    def masked_loss(self, x, t, padding_mask, mask_in, mask_pred):
        """MSE on the hidden points given explicit masks (what masked_pred + nn.MSELoss compute, ref :183-231)."""
        x_masked = x.clone()
        x_masked[~mask_in] = 0
        return masked_mse(self(x_masked, t, mask=padding_mask), x, mask_pred)

    def _step(self, batch, name):
        if len(batch) == 3:
            t, x, padding_mask = batch
        else:
            _, x, t, padding_mask, *_ = batch
        # masks are built on the padding mask's device; only the per-band counts of observed points (B x nband integers)
        # come to the host, for the bounds of the random run starts
        mask_in, mask_pred = get_continous_random_mask(padding_mask.to(x.device), self.nband, f_mask=self.f_mask)
        loss = self.masked_loss(x, t, padding_mask, mask_in, mask_pred)
        self.log(name, loss, on_epoch=True, on_step=False, prog_bar=True)
        return loss

    def training_step(self, batch, batch_idx):
        return self._step(batch, "train_loss")

    def validation_step(self, batch, batch_idx):
        return self._step(batch, "val_loss")

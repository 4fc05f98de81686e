"""The CLIP-style module with the reference's surface (src/models_multimodal.py): class names,
constructor arguments, encoder slots, method names, the 9-tuple batch and the state_dict keys
are the reference's; the arithmetic is libmsn_hip's.  torch.nn.Conv2d / BatchNorm2d / Linear /
Embedding objects are parameter holders only (their forward is never called).

Out of scope here: the supervised `regression` / `classification` heads of
the reference constructor -- requesting them raises NotImplementedError.
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import functional as F_
from . import ops
from . import markers
from .loss import clip_loss_multimodal
from .transformer_utils import TransformerWithTimeEmbeddings


class Residual(nn.Module):
    """fn(x) + x -- ref src/models_multimodal.py:24-35 (kept for the `...0.fn.*` state_dict keys)."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class ConvMixer(nn.Module):
    """ref src/models_multimodal.py:38-95.  (B, C, H, W) -> (B, n_out)."""

    def __init__(self, dim, depth, channels=1, kernel_size=5, patch_size=8, n_out=128, dropout_prob=0.5):
        super().__init__()
        self.dim, self.depth, self.patch_size = dim, depth, patch_size
        self.net = nn.Sequential(nn.Conv2d(channels, dim, kernel_size=patch_size, stride=patch_size, bias=False),
                                 nn.GELU(), nn.BatchNorm2d(dim))
        for _ in range(depth):
            self.net.append(nn.Sequential(
                Residual(nn.Sequential(nn.Conv2d(dim, dim, kernel_size, groups=dim, padding="same"), nn.GELU(),
                                       nn.BatchNorm2d(dim), nn.Dropout(dropout_prob))),
                nn.Conv2d(dim, dim, kernel_size=1), nn.GELU(), nn.BatchNorm2d(dim), nn.Dropout(dropout_prob)))
        self.projection = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Linear(dim, 1024), nn.GELU(),
                                        nn.Dropout(dropout_prob), nn.Linear(1024, n_out))
        self._dropout = dropout_prob

    @staticmethod
    def _bn(bn):
        return [bn.weight, bn.bias, bn.running_mean, bn.running_var]

    def _flat_params(self):
        flat = [self.net[0].weight] + self._bn(self.net[2])
        for i in range(self.depth):
            layer = self.net[3 + i]
            dw, bn_a = layer[0].fn[0], layer[0].fn[2]
            flat += [dw.weight, dw.bias] + self._bn(bn_a) + [layer[1].weight, layer[1].bias] + self._bn(layer[3])
        return flat

    def _bns(self):
        yield self.net[2]
        for i in range(self.depth):
            yield self.net[3 + i][0].fn[2]
            yield self.net[3 + i][3]

    def forward(self, x):
        p_drop = self._dropout if self.training else 0.0
        tokens = F_.convmixer_trunk(x, self.training, self.depth, self.patch_size, self._flat_params(), drop_p=p_drop)
        if self.training:
            with torch.no_grad():
                for bn in self._bns():
                    bn.num_batches_tracked += 1
        B, hw, _ = tokens.shape
        ones = torch.ones((B, hw), dtype=torch.uint8, device=tokens.device)
        pooled = F_.masked_pool(tokens, ones, "mean")                      # AdaptiveAvgPool2d((1, 1)) + Flatten
        p2, p5 = self.projection[2], self.projection[5]
        return F_.linear_chain(pooled, [(p2.weight, p2.bias), (p5.weight, p5.bias)], [F_.ACT_GELU, F_.ACT_NONE],
                               drop_p=p_drop)


class MLP(nn.Module):
    """ref src/models_multimodal.py:834-856: num_layers x (Linear, ReLU, Dropout), then Linear
    (state_dict keys layers.0, layers.3, layers.6, ...)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, dropout):
        super().__init__()
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.num_layers, self.dropout = num_layers, dropout
        self.layers = nn.ModuleList()
        self.layers.append(nn.Linear(input_dim, hidden_dim))
        self.layers.append(nn.ReLU())
        self.layers.append(nn.Dropout(dropout))
        for _ in range(num_layers - 1):
            self.layers.append(nn.Linear(hidden_dim, hidden_dim))
            self.layers.append(nn.ReLU())
            self.layers.append(nn.Dropout(dropout))
        self.layers.append(nn.Linear(hidden_dim, output_dim))

    def forward(self, x):
        lin = [m for m in self.layers if isinstance(m, nn.Linear)]
        return F_.linear_chain(x, [(m.weight, m.bias) for m in lin], [F_.ACT_RELU] * (len(lin) - 1) + [F_.ACT_NONE],
                               drop_p=self.dropout if self.training else 0.0)


class LightCurveImageCLIP(nn.Module):
    """ref src/models_multimodal.py:98-556 (contrastive branch).  A plain nn.Module exposing the
    LightningModule hooks the reference implements (training_step, validation_step,
    configure_optimizers, on_*), driven by multimodal_supernovae_amd.trainer.Trainer."""

    def __init__(self, enc_dim: int = 128, logit_scale: float = 10.0, nband: int = 1,
                 transformer_kwargs: Optional[Dict] = None, transformer_spectral_kwargs: Optional[Dict] = None,
                 conv_kwargs: Optional[Dict] = None, meta_kwargs: Optional[Dict] = None,
                 combinations: List[str] = ("host_galaxy", "spectral"), optimizer_kwargs: Optional[Dict] = None,
                 lr: float = 1e-4, loss: str = "sigmoid", regression: bool = False, classification: bool = False,
                 n_classes: int = 5, global_negatives: bool = True):
        super().__init__()
        if regression or classification:
            raise NotImplementedError("the supervised regression / classification heads are outside the "
                                      "contrastive hot path this package implements")
        default_t = {"n_out": 128, "emb": 256, "heads": 2, "depth": 8, "time_norm": 10000.0}
        transformer_kwargs = dict(transformer_kwargs or default_t)
        transformer_spectral_kwargs = dict(transformer_spectral_kwargs or default_t)
        conv_kwargs = dict(conv_kwargs or {"dim": 32, "depth": 8, "channels": 3, "kernel_size": 5,
                                           "patch_size": 10, "n_out": 128})
        meta_kwargs = dict(meta_kwargs or {"input_dim": 128, "hidden_dim": 128, "num_layers": 2})
        self.lr = lr
        self.optimizer_kwargs = dict(optimizer_kwargs or {})
        self.enc_dim = enc_dim
        self.combinations = set(combinations)
        self.regression, self.classification = False, False
        self.global_negatives = global_negatives
        self.logit_scale = nn.Parameter(torch.tensor(math.log(logit_scale)), requires_grad=True)
        self.logit_bias = nn.Parameter(torch.tensor(-10.0), requires_grad=True)
        if "lightcurve" in self.combinations:
            self.lightcurve_encoder = TransformerWithTimeEmbeddings(nband=nband, **transformer_kwargs)
            self.lightcurve_projection = nn.Linear(transformer_kwargs["n_out"], enc_dim)
        if "spectral" in self.combinations:
            self.spectral_encoder = TransformerWithTimeEmbeddings(nband=1, **transformer_spectral_kwargs)
            self.spectral_projection = nn.Linear(transformer_spectral_kwargs["n_out"], enc_dim)
        if "host_galaxy" in self.combinations:
            self.image_encoder = ConvMixer(**conv_kwargs)
            self.image_projection = nn.Linear(conv_kwargs["n_out"], enc_dim)
        if "meta" in self.combinations:
            self.len_meta_input = meta_kwargs["input_dim"]
            self.class_emb = nn.Embedding(n_classes, self.len_meta_input // 2)
            meta_kwargs.setdefault("dropout", 0.0)
            self.meta_encoder = MLP(output_dim=enc_dim, **meta_kwargs)
        self.loss = loss
        self.embs_list = None
        self.logged = {}

    # -- logging stand-in for LightningModule.log: last value per key, readable by the trainer --
    def log(self, name, value, **kwargs):
        # detached: a logged loss must not keep its autograd graph (and the AccumulateGrad nodes of every parameter,
        # bound to the stream of that step) alive into the next step
        self.logged[name] = value.detach() if torch.is_tensor(value) else value

    # -- forward: list of unit-norm embeddings in the FIXED order img, lc, sp, meta (ref :259-273) --
    # Towers are independent until the loss: every tower after the first is enqueued on its own HIP stream, so the
    # vector-ALU / HBM-bound kernels of a light-curve or spectrum tower fill the CUs a matrix-core GEMM of the image
    # tower leaves idle (round tails, launch gaps).  autograd replays each tower's backward on the stream its forward
    # ran on, with the joins it needs; results are unchanged (no cross-stream reductions).
    concurrent_towers = True

    def _side_streams(self, n, device):
        pool = getattr(self, "_tower_streams", None)
        if pool is None or len(pool) < n or pool[0].device != device:
            pool = self._tower_streams = [torch.cuda.Stream(device=device) for _ in range(n)]
        return pool[:n]

    def forward(self, x_img, x_lc, t_lc, mask_lc, x_sp, t_sp, mask_sp, redshift=None, classification=None):
        towers = []
        if "host_galaxy" in self.combinations:
            towers.append(lambda: self.image_embeddings_with_projection(x_img))
        if "lightcurve" in self.combinations:
            towers.append(lambda: self.lightcurve_embeddings_with_projection(x_lc, t_lc, mask_lc))
        if "spectral" in self.combinations:
            towers.append(lambda: self.spectral_embeddings_with_projection(x_sp, t_sp, mask_sp))
        if "meta" in self.combinations:
            towers.append(lambda: self.meta_embeddings_with_projection(classification, redshift))
        device = self.logit_scale.device
        if not (self.concurrent_towers and device.type == "cuda" and len(towers) > 1):
            return [t() for t in towers]
        main = torch.cuda.current_stream(device)
        side = self._side_streams(len(towers) - 1, device)
        out = [None] * len(towers)
        for i in range(1, len(towers)):           # the later (smaller) towers first: their queues fill while the
            side[i - 1].wait_stream(main)         # first tower's launches follow on the caller's stream
            with torch.cuda.stream(side[i - 1]):
                out[i] = towers[i]()
        out[0] = towers[0]()
        for i in range(1, len(towers)):
            main.wait_stream(side[i - 1])
            out[i].record_stream(main)
        return out

    def image_embeddings_with_projection(self, x_img):
        with markers.range("image tower forward"):
            h = self.image_encoder(x_img)
            return F_.project_normalise(h, self.image_projection.weight, self.image_projection.bias)

    def lightcurve_embeddings_with_projection(self, x_lc, t_lc, mask_lc=None):
        with markers.range("light-curve tower forward"):
            h = self.lightcurve_encoder(x_lc[..., None], t_lc, mask_lc)
            return F_.project_normalise(h, self.lightcurve_projection.weight, self.lightcurve_projection.bias)

    def spectral_embeddings_with_projection(self, x_lc, t_lc, mask_lc=None):
        with markers.range("spectrum tower forward"):
            h = self.spectral_encoder(x_lc[..., None], t_lc, mask_lc)
            return F_.project_normalise(h, self.spectral_projection.weight, self.spectral_projection.bias)

    def meta_embeddings_with_projection(self, classification, redshift):
        half = self.len_meta_input // 2
        # gather + broadcast + concat are index plumbing (ref :296-302); the MLP runs on the kernels
        x_meta = torch.cat([self.class_emb.weight[classification.long()], redshift.float()[:, None].repeat(1, half)],
                           dim=-1)
        return F_.l2_normalise(self.meta_encoder(x_meta))

    def configure_optimizers(self):
        from .optim import RAdam
        return {"optimizer": RAdam(self.parameters(), lr=self.lr, **self.optimizer_kwargs)}

    def _loss(self, embs):
        if self.loss == "softmax":
            with markers.range("InfoNCE forward (+ exchange)"):
                return clip_loss_multimodal(embs, self.logit_scale, self.logit_bias,
                                            global_negatives=self.global_negatives)
        if self.loss == "sigmoid":
            from .loss import sigmoid_loss_multimodal
            return sigmoid_loss_multimodal(embs, self.logit_scale, self.logit_bias,
                                           global_negatives=self.global_negatives)
        raise ValueError(f"unknown loss {self.loss!r}")

    def training_step(self, batch, batch_idx):
        embs = self(*batch)
        loss = self._loss(embs)
        self.log("train_loss", loss, on_epoch=True, on_step=False, prog_bar=True, logger=True)
        return loss

    def on_train_epoch_start(self):
        pass

    def on_train_epoch_end(self):
        pass

    def on_validation_start(self):
        self.embs_list = [[] for _ in range(len(self.combinations))]

    def validation_step(self, batch, batch_idx):
        embs = self(*batch)
        for i, e in enumerate(embs):
            self.embs_list[i].append(e.detach())
        loss = self._loss(embs)
        self.log("val_loss", loss, on_epoch=True, on_step=False, prog_bar=True, logger=True)
        return loss

    def on_validation_epoch_end(self):
        """ref :519-556: concatenate the stored embeddings and log the retrieval AUC per modality pair."""
        from .utils import get_AUC
        if not self.embs_list or not self.embs_list[0]:
            self.embs_list = None
            return
        embs = [torch.cat(e, dim=0) for e in self.embs_list]
        if len(embs) == 2:
            self.log("AUC_val", get_AUC(embs[0], embs[1]), on_epoch=True, on_step=False, prog_bar=True, logger=True)
        else:
            count = 1
            for i in range(len(embs) - 1):
                for j in range(i + 1, len(embs)):
                    self.log(f"AUC_val{count}", get_AUC(embs[i], embs[j]), on_epoch=True, on_step=False, prog_bar=True,
                             logger=True)
                    count += 1
        self.embs_list = None

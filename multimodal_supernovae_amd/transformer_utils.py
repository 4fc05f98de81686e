"""Light-curve / spectrum encoders with the reference's class names, constructor arguments and
state_dict keys (src/transformer_utils.py), computing through libmsn_hip only.

torch.nn.Linear / LayerNorm / Embedding / MultiheadAttention objects appear here purely as
PARAMETER HOLDERS (same names, shapes and default initialisation as the reference, so its
checkpoints load with strict=True); their own forward methods are never called.
"""
import math

import torch
import torch.nn as nn

from . import functional as F_
from . import ops


def _mask_bytes(mask):
    return None if mask is None else ops._mask_u8(mask)


class SelfAttention(nn.Module):
    """Multi-head self-attention, ref src/transformer_utils.py:8-89 (scores / sqrt(emb), -1e7 key fill)."""

    def __init__(self, emb, heads=2):
        super().__init__()
        assert emb % heads == 0, f"Embedding dimension ({emb}) should be divisible by nr. of heads ({heads})"
        self.emb, self.heads = emb, heads
        self.tokeys = nn.Linear(emb, emb, bias=False)
        self.toqueries = nn.Linear(emb, emb, bias=False)
        self.tovalues = nn.Linear(emb, emb, bias=False)
        self.unifyheads = nn.Linear(emb, emb)
        # the three projection weights share one buffer once stacked_qkv() has run: a state_dict handed to
        # safetensors.save_file / torch.save must not expose three views of one storage (safetensors refuses them,
        # torch.save would write the 3e x e storage per key) -- every entry of this module is saved as its own copy
        self._register_state_dict_hook(SelfAttention._unshare_saved_weights)

    @staticmethod
    def _unshare_saved_weights(module, state_dict, prefix, local_metadata):
        for name in ("toqueries", "tokeys", "tovalues"):
            key = f"{prefix}{name}.weight"
            t = state_dict.get(key)
            if t is not None and t.untyped_storage().nbytes() != t.numel() * t.element_size():
                state_dict[key] = t.clone()
        return state_dict

    def stacked_qkv(self):
        """The (3 emb, emb) matrix [toqueries ; tokeys ; tovalues] the fused projection multiplies by, WITHOUT a copy per
        step: the three weights live side by side in one buffer (the parameters are re-pointed at row blocks of it the
        first time -- and again should a `.to()` / `.cuda()` have moved them apart), so the optimiser's in-place updates
        land in the stacked matrix directly.  state_dict keys, shapes and values are unchanged."""
        wq, wk, wv = self.toqueries.weight, self.tokeys.weight, self.tovalues.weight
        e = self.emb
        nbytes = e * e * wq.element_size()
        adjacent = (wq.is_contiguous() and wk.is_contiguous() and wv.is_contiguous() and wq.device == wk.device == wv.device
                    and wk.data_ptr() == wq.data_ptr() + nbytes and wv.data_ptr() == wk.data_ptr() + nbytes
                    and wq.untyped_storage().data_ptr() == wv.untyped_storage().data_ptr())
        if not adjacent:
            if torch.cuda.is_available() and wq.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("SelfAttention weights must be stacked before a HIP-graph capture: run one eager step first")
            with torch.no_grad():
                stacked = torch.cat([wq.detach(), wk.detach(), wv.detach()], dim=0)
                wq.data, wk.data, wv.data = stacked[:e], stacked[e:2 * e], stacked[2 * e:]
        return wq.data.as_strided((3 * e, e), (e, 1))

    def forward(self, x, mask=None):
        assert x.shape[-1] == self.emb, f"Input embedding dim ({x.shape[-1]}) should match layer embedding dim ({self.emb})"
        return F_.self_attention(x, mask, self.heads, self.tokeys.weight, self.toqueries.weight, self.tovalues.weight,
                                 self.unifyheads.weight, self.unifyheads.bias, wcat=self.stacked_qkv())


class TransformerBlock(nn.Module):
    """Post-norm block with a ReLU feed-forward, ref src/transformer_utils.py:92-116."""

    def __init__(self, emb, heads, ff_hidden_mult=6, dropout=0.0):
        super().__init__()
        self.attention = SelfAttention(emb, heads=heads)
        self.norm1 = nn.LayerNorm(emb)
        self.norm2 = nn.LayerNorm(emb)
        self.ff = nn.Sequential(nn.Linear(emb, ff_hidden_mult * emb), nn.ReLU(), nn.Linear(ff_hidden_mult * emb, emb))
        self.do = nn.Dropout(dropout)     # holder of p: the dropout itself runs in the fused block (train mode only)

    def _params(self):
        a = self.attention
        return {"tokeys": a.tokeys.weight, "toqueries": a.toqueries.weight, "tovalues": a.tovalues.weight,
                "unify_w": a.unifyheads.weight, "unify_b": a.unifyheads.bias,
                "norm1_w": self.norm1.weight, "norm1_b": self.norm1.bias,
                "ff0_w": self.ff[0].weight, "ff0_b": self.ff[0].bias, "ff2_w": self.ff[2].weight,
                "ff2_b": self.ff[2].bias, "norm2_w": self.norm2.weight, "norm2_b": self.norm2.bias}

    def forward(self, x, mask=None, ff_planes=None):
        return F_.post_norm_block(x, _mask_bytes(mask) if mask is None or mask.dtype != torch.uint8 else mask,
                                  self.attention.heads, self._params(), drop_p=self.do.p if self.training else 0.0,
                                  wcat=self.attention.stacked_qkv(), ff_planes=ff_planes)


class Transformer(nn.Module):
    """`depth` blocks, no final norm -- ref src/transformer_utils.py:119-153 (ff_hidden_mult = 4)."""

    def __init__(self, emb, heads, depth, ff_hidden_mult=4, dropout=0.0):
        super().__init__()
        self.tblocks = nn.ModuleList(
            [TransformerBlock(emb=emb, heads=heads, ff_hidden_mult=ff_hidden_mult, dropout=dropout) for _ in range(depth)])
        self.do = nn.Dropout(dropout)

    def forward(self, x, mask=None):
        m = _mask_bytes(mask)
        x = F_.dropout(x, self.do.p, self.training)           # ref :150
        planes = None
        if len(self.tblocks) and F_.fused_ff_applies(x, x.shape[-1], self.tblocks[0].ff[0].weight.shape[0]):
            # the fused feed-forward kernels keep both weights in LDS as plane matrices: ONE launch splits those of every block
            mats = [w for blk in self.tblocks for w in (blk.ff[0].weight, blk.ff[2].weight)]
            planes = ops.plane_split_list(mats, 3, transposed=[False, True] * len(self.tblocks))
        for i, blk in enumerate(self.tblocks):
            x = blk(x, m, ff_planes=None if planes is None else (planes[2 * i], planes[2 * i + 1]))
        return x


class TimePositionalEncoding(nn.Module):
    """Interleaved sin / cos of t * norm^(-2k/d) -- ref src/transformer_utils.py:156-176."""

    def __init__(self, d_emb, norm=10000.0):
        super().__init__()
        self.d_emb, self.norm = d_emb, norm
        # the frequency table is computed exactly as the reference does (fp32 on the host, :169-171)
        omega = torch.exp(torch.arange(0, d_emb, 2).float() * (-math.log(norm) / d_emb))
        self.register_buffer("_omega", omega, persistent=False)

    def forward(self, t):
        zeros_e = torch.zeros(self.d_emb, dtype=torch.float32, device=t.device)
        return ops.time_embed_fwd(torch.zeros_like(t, dtype=torch.float32), t.float().contiguous(), zeros_e, zeros_e,
                                  self._omega.to(t.device))


class TransformerWithTimeEmbeddings(nn.Module):
    """ref src/transformer_utils.py:179-253: value + time (+ band) embedding -> Transformer ->
    zero the padded tokens -> mean | max | attn pooling -> projection.  `agg="pretraining"`
    returns the zeroed tokens (B, T, emb)."""

    def __init__(self, n_out, nband=1, agg="mean", time_norm=10000.0, **kwargs):
        super().__init__()
        emb = kwargs["emb"]
        self.agg, self.nband = agg, nband
        self.embedding_mag = nn.Linear(in_features=1, out_features=emb)
        self.embedding_t = TimePositionalEncoding(emb, time_norm)
        self.transformer = Transformer(**kwargs)
        if nband > 1:
            self.band_emb = nn.Embedding(nband, emb)
        self.projection = nn.Linear(emb, n_out)
        if self.agg == "attn":
            self.query = nn.Parameter(torch.rand(emb))
            self.agg_attn = nn.MultiheadAttention(embed_dim=emb, num_heads=2, dropout=0.0, batch_first=True)

    def forward(self, x, t, mask=None):
        if mask is None:
            raise TypeError("TransformerWithTimeEmbeddings needs the (B, T) padding mask (the reference "
                            "multiplies by it unconditionally, src/transformer_utils.py:235)")
        B, T = t.shape
        if self.nband > 1 and T % self.nband != 0:
            raise RuntimeError(f"sequence length {T} is not divisible by nband={self.nband}")
        m = _mask_bytes(mask)
        h = F_.time_embed(x.reshape(B, T).float(), t.float(), self.embedding_mag.weight, self.embedding_mag.bias,
                          self.band_emb.weight if self.nband > 1 else None, self.embedding_t._omega)
        h = self.transformer(h, m)
        if self.agg in ("mean", "max"):
            h = F_.masked_pool(h, m, self.agg)       # zeroing + reduction in one kernel
        else:
            h = F_.mask_tokens(h, m)
            if self.agg == "attn":
                a = self.agg_attn
                h = F_.attn_pool(h, self.query, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias)
            elif self.agg == "pretraining":
                return h
        return F_.linear(h, self.projection.weight, self.projection.bias)

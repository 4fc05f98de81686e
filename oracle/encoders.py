"""Oracle: reference encoders, functional form (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Every function takes a flat parameter mapping ``P`` keyed by the reference's
state_dict names plus a ``prefix`` (e.g. ``"lightcurve_encoder."``) and restates
the arithmetic of /root/reference/src/transformer_utils.py and the ConvMixer / MLP
classes of /root/reference/src/models_multimodal.py with einsum-level torch ops.
Pinned against the reference by tests/golden/{attn,block,tenc,convmixer,mlp}_*.npz.
"""
import math

import torch
import torch.nn.functional as F

MASK_FILL = -1e7  # ref transformer_utils.py:77 (a finite fill, not -inf)


def linear(P, name, x, bias=True):
    y = x @ P[name + ".weight"].T
    return y + P[name + ".bias"] if bias else y


def self_attention(P, prefix, x, mask, heads):
    """ref transformer_utils.py:36-89 (SelfAttention.forward).

    q, k, v: bias-free projections (:45-47); scores divided by sqrt(emb) -- the FULL
    embedding width, not the head width (:63-64); key-padding positions are *replaced*
    by -1e7 (:73-77); softmax over keys (:79); heads merged then `unifyheads` (+bias) (:84-89).
    """
    b, t, e = x.shape
    s = e // heads
    q = linear(P, prefix + "toqueries", x, bias=False).view(b, t, heads, s)
    k = linear(P, prefix + "tokeys", x, bias=False).view(b, t, heads, s)
    v = linear(P, prefix + "tovalues", x, bias=False).view(b, t, heads, s)
    quarter = e ** 0.25
    dot = torch.einsum("bihs,bjhs->bhij", q / quarter, k / quarter)
    if mask is not None:
        keep = mask.to(torch.bool)[:, None, None, :]
        dot = torch.where(keep, dot, torch.full_like(dot, MASK_FILL))
    p = torch.softmax(dot, dim=-1)
    out = torch.einsum("bhij,bjhs->bihs", p, v).reshape(b, t, e)
    return linear(P, prefix + "unifyheads", out)


def layer_norm(P, name, x, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * P[name + ".weight"] + P[name + ".bias"]


def transformer_block(P, prefix, x, mask, heads):
    """Post-norm block, ReLU feed-forward -- ref transformer_utils.py:109-116 (dropout p=0)."""
    x = layer_norm(P, prefix + "norm1", self_attention(P, prefix + "attention.", x, mask, heads) + x)
    h = torch.relu(linear(P, prefix + "ff.0", x))
    return layer_norm(P, prefix + "norm2", linear(P, prefix + "ff.2", h) + x)


def transformer(P, prefix, x, mask, heads, depth):
    """ref transformer_utils.py:143-153: `depth` blocks, no final norm."""
    for i in range(depth):
        x = transformer_block(P, f"{prefix}tblocks.{i}.", x, mask, heads)
    return x


def time_positional_encoding(t, emb, norm):
    """ref transformer_utils.py:166-176: interleaved sin/cos of t * norm^(-2k/emb)."""
    k2 = torch.arange(0, emb, 2, dtype=torch.float32, device=t.device)
    omega = torch.exp(k2 * (-math.log(norm) / emb))
    ang = t[:, :, None] * omega
    return torch.stack((torch.sin(ang), torch.cos(ang)), dim=-1).reshape(t.shape[0], t.shape[1], emb)


def attn_pool(P, prefix, x):
    """ref transformer_utils.py:240-246: one learnable query through nn.MultiheadAttention
    (2 heads, batch_first, NO key-padding mask: zeroed padded tokens still take part)."""
    b, t, e = x.shape
    hd = e // 2
    w, bias = P[prefix + "agg_attn.in_proj_weight"], P[prefix + "agg_attn.in_proj_bias"]
    q = (P[prefix + "query"] @ w[:e].T + bias[:e]).view(2, hd)
    k = (x @ w[e:2 * e].T + bias[e:2 * e]).view(b, t, 2, hd)
    v = (x @ w[2 * e:].T + bias[2 * e:]).view(b, t, 2, hd)
    p = torch.softmax(torch.einsum("hs,bjhs->bhj", q, k) / math.sqrt(hd), dim=-1)
    o = torch.einsum("bhj,bjhs->bhs", p, v).reshape(b, e)
    return o @ P[prefix + "agg_attn.out_proj.weight"].T + P[prefix + "agg_attn.out_proj.bias"]


def transformer_with_time_embeddings(P, prefix, x, t, mask, *, emb, heads, depth, time_norm,
                                     nband=1, agg="mean"):
    """ref transformer_utils.py:209-253.

    x: (B, T, 1) values, t: (B, T) times, mask: (B, T) bool (mandatory, as in the reference).
    """
    h = x * P[prefix + "embedding_mag.weight"][:, 0] + P[prefix + "embedding_mag.bias"]
    h = h + time_positional_encoding(t, emb, time_norm)
    if nband > 1:
        band = torch.arange(nband, device=t.device).repeat_interleave(x.shape[1] // nband)
        h = h + P[prefix + "band_emb.weight"][band][None]
    h = transformer(P, prefix + "transformer.", h, mask, heads, depth)
    h = h * mask[:, :, None]
    if agg == "mean":
        h = h.sum(dim=1) / mask.sum(dim=1)[:, None]
    elif agg == "max":
        h = h.max(dim=1)[0]
    elif agg == "attn":
        h = attn_pool(P, prefix, h)
    elif agg == "pretraining":
        return h
    return linear(P, prefix + "projection", h)


def _bn(P, name, x, training, stats_out=None, momentum=0.1, eps=1e-5):
    """BatchNorm2d over (B, C, H, W); train mode = batch statistics (biased var for the
    normalisation, unbiased for the running estimate), eval = running statistics."""
    if training:
        mu = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if stats_out is not None:
            n = x.numel() // x.shape[1]
            stats_out[name + ".running_mean"] = (
                (1 - momentum) * P[name + ".running_mean"] + momentum * mu.detach())
            stats_out[name + ".running_var"] = (
                (1 - momentum) * P[name + ".running_var"] + momentum * var.detach() * n / (n - 1))
    else:
        mu, var = P[name + ".running_mean"], P[name + ".running_var"]
    xh = (x - mu[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
    return xh * P[name + ".weight"][None, :, None, None] + P[name + ".bias"][None, :, None, None]


def convmixer(P, prefix, x, *, depth, patch_size, training=True, stats_out=None):
    """ref models_multimodal.py:52-95 (ConvMixer) + :24-35 (Residual), dropout p=0.

    stem: bias-free patch conv -> GELU -> BN (:53-59); per layer: x + BN(GELU(depthwise 'same'))
    then BN(GELU(1x1 conv)) (:62-79); head: global mean -> Linear(dim,1024) -> GELU -> Linear (:82-89).
    """
    net = prefix + "net."
    x = F.conv2d(x, P[net + "0.weight"], None, stride=patch_size)
    x = _bn(P, net + "2", F.gelu(x), training, stats_out)
    for i in range(depth):
        blk = f"{net}{3 + i}."
        dw = F.conv2d(x, P[blk + "0.fn.0.weight"], P[blk + "0.fn.0.bias"], padding="same",
                      groups=x.shape[1])
        x = _bn(P, blk + "0.fn.2", F.gelu(dw), training, stats_out) + x
        x = F.conv2d(x, P[blk + "1.weight"], P[blk + "1.bias"])
        x = _bn(P, blk + "3", F.gelu(x), training, stats_out)
    x = x.mean(dim=(2, 3))
    x = F.gelu(linear(P, prefix + "projection.2", x))
    return linear(P, prefix + "projection.5", x)


def mlp(P, prefix, x, num_layers):
    """ref models_multimodal.py:834-856: num_layers x (Linear, ReLU, Dropout) then Linear;
    the Linear modules sit at indices 0, 3, 6, ... of `layers`."""
    for i in range(num_layers):
        x = torch.relu(linear(P, f"{prefix}layers.{3 * i}", x))
    return linear(P, f"{prefix}layers.{3 * num_layers}", x)

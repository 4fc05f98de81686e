"""Oracle: contrastive losses (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates /root/reference/src/loss.py in closed form (log-sum-exp minus diagonal)
instead of the reference's LogSoftmax(...).diag() pipeline; pinned against the
reference by tests/golden/loss_*.npz.
"""
from itertools import combinations as _pairs

import torch
import torch.nn.functional as F


def _per_pair(value, n_pairs):
    """0-dim scale/bias is shared by every modality pair (ref loss.py:49-52, :90-93)."""
    return value.repeat(n_pairs) if value.dim() == 0 else value


def clip_loss(embs1, embs2, logit_scale, logit_bias):
    """Symmetric softmax InfoNCE -- ref src/loss.py:14-38.

    logits S = (embs2 . embs1^T) * exp(logit_scale) + logit_bias          (:22-24)
    loss = 1/2 * [ sum_i (LSE_row_i - S_ii) + sum_i (LSE_col_i - S_ii) ] / n,
    n = min(len(embs1), len(embs2)); the diagonal has n entries      (:26-37)
    """
    s = torch.exp(logit_scale)
    S = (embs2 @ embs1.T) * s + logit_bias
    n = min(embs1.shape[0], embs2.shape[0])
    d = torch.diagonal(S)
    row = torch.logsumexp(S, dim=1)[:n]
    col = torch.logsumexp(S, dim=0)[:n]
    return 0.5 * ((row - d).sum() / n + (col - d).sum() / n)


def clip_loss_multimodal(embeddings, logit_scales, logit_biases):
    """Sum (not mean) of clip_loss over modality pairs i<j -- ref src/loss.py:41-65."""
    m = len(embeddings)
    n_pairs = m * (m - 1) // 2
    scales = _per_pair(logit_scales, n_pairs)
    biases = _per_pair(logit_biases, n_pairs)
    total = 0
    for k, (i, j) in enumerate(_pairs(range(m), 2)):
        total = total + clip_loss(embeddings[i], embeddings[j], scales[k], biases[k])
    return total


def sigmoid_loss(embs1, embs2, logit_scale, logit_bias):
    """SigLIP-style loss with the reference's own sign convention -- ref src/loss.py:68-83.

    labels z = 2I - 1; logits Z = -(embs2 . embs1^T) * exp(scale) + bias, cast to fp64 (:78-79);
    loss = -mean(log sigmoid(-z * Z)) over all bs*bs entries (:81).
    logsigmoid is used for the log(sigmoid(.)) composition (same value, no -inf at
    large arguments).
    """
    bs = embs2.shape[0]
    z = 2.0 * torch.eye(bs, device=embs2.device) - torch.ones(bs, bs, device=embs2.device)
    Z = (-(embs2 @ embs1.T) * torch.exp(logit_scale) + logit_bias).to(torch.float64)
    return -F.logsigmoid(-z.to(torch.float64) * Z).mean()


def sigmoid_loss_multimodal(embeds, logit_scales, logit_biases):
    """Pairwise sum of sigmoid_loss -- ref src/loss.py:86-107."""
    m = len(embeds)
    n_pairs = m * (m - 1) // 2
    scales = _per_pair(logit_scales, n_pairs)
    biases = _per_pair(logit_biases, n_pairs)
    total = 0
    for k, (i, j) in enumerate(_pairs(range(m), 2)):
        total = total + sigmoid_loss(embeds[i], embeds[j], scales[k], biases[k])
    return total

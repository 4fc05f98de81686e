"""Contrastive losses with the reference's signatures (src/loss.py) on the fused HIP kernels.

    clip_loss(embs1, embs2, logit_scale, logit_bias)               -- ref src/loss.py:14-38
    clip_loss_multimodal(embeddings, logit_scales, logit_biases)   -- ref src/loss.py:41-65

`logit_scale` is the LOG of the scale (the kernel exponentiates it, ref :22).  Tensors in,
0-dim tensor out, differentiable w.r.t. both embedding matrices, the scale and the bias.

Data parallel ("global negatives"): when torch.distributed is initialised with world_size > 1
and `global_negatives=True`, every rank passes its LOCAL rows; the embeddings (and afterwards the
per-row log-sum-exps, 2N floats) are all-gathered over RCCL, each rank evaluates only its own
rows / columns of the logit matrix, and the returned loss is the all-reduced global value.  The
gradient each rank gets for its local rows is already the full gradient of the global loss, so
parameter gradients must be SUMMED over ranks (see distributed.py), not averaged.
"""
from itertools import combinations as _pairs

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


class HipPairKernels:
    """The product compute backend: msn_infonce_fwd / msn_infonce_bwd through the C-ABI."""

    @staticmethod
    def forward(e1_loc, e2_loc, e1_all, e2_all, q_offset, log_scale, bias):
        _lib.require_gpu()
        b1, D = e1_loc.shape
        b2 = e2_loc.shape[0]
        n1, n2 = e1_all.shape[0], e2_all.shape[0]
        dev = e1_loc.device
        L = lib()
        nb = L.msn_infonce_workspace_bytes(b1, b2, n1, n2, D)
        ws = torch.empty(max(nb, 16) // 4 + 1, dtype=torch.float32, device=dev)
        lse_row = torch.empty(b2, dtype=torch.float32, device=dev)
        lse_col = torch.empty(b1, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        check(L.msn_infonce_fwd(ptr(e1_loc), e1_loc.stride(0), b1, ptr(e2_loc), e2_loc.stride(0), b2,
                                ptr(e1_all), e1_all.stride(0), n1, ptr(e2_all), e2_all.stride(0), n2,
                                D, q_offset, ptr(log_scale), ptr(bias), ptr(lse_row), ptr(lse_col), ptr(loss),
                                ptr(ws), nb, stream_ptr()), "msn_infonce_fwd")
        return lse_row, lse_col, loss

    @staticmethod
    def backward(e1_loc, e2_loc, e1_all, e2_all, q_offset, log_scale, bias, lse_row_all, lse_col_all, grad_out):
        b1, D = e1_loc.shape
        b2 = e2_loc.shape[0]
        n1, n2 = e1_all.shape[0], e2_all.shape[0]
        dev = e1_loc.device
        L = lib()
        nb = L.msn_infonce_workspace_bytes(b1, b2, n1, n2, D)
        ws = torch.empty(max(nb, 16) // 4 + 1, dtype=torch.float32, device=dev)
        d1 = torch.empty((b1, D), dtype=torch.float32, device=dev)
        d2 = torch.empty((b2, D), dtype=torch.float32, device=dev)
        dsb = torch.empty(2, dtype=torch.float32, device=dev)
        check(L.msn_infonce_bwd(ptr(e1_loc), e1_loc.stride(0), b1, ptr(e2_loc), e2_loc.stride(0), b2,
                                ptr(e1_all), e1_all.stride(0), n1, ptr(e2_all), e2_all.stride(0), n2,
                                D, q_offset, ptr(log_scale), ptr(bias), ptr(lse_row_all), ptr(lse_col_all),
                                ptr(grad_out), ptr(d1), D, ptr(d2), D, ptr(dsb), ptr(ws), nb, stream_ptr()),
              "msn_infonce_bwd")
        return d1, d2, dsb[0], dsb[1]


def _gather_rows(t, group):
    """All-gather equal-sized row blocks -> (world * b, ...) in rank order (no autograd)."""
    world = dist.get_world_size(group)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out


class _PairLoss(torch.autograd.Function):
    """One modality pair; single process or row-sharded over `group`."""

    @staticmethod
    def forward(ctx, e1, e2, log_scale, bias, kernels, group, sharded):
        e1 = e1.contiguous()
        e2 = e2.contiguous()
        log_scale = log_scale.detach().to(torch.float32).reshape(()).contiguous()
        bias = bias.detach().to(torch.float32).reshape(()).contiguous()
        if sharded:
            if e1.shape[0] != e2.shape[0]:
                raise ValueError("global-negatives mode needs the same local batch for both modalities")
            rank = dist.get_rank(group)
            e1_all, e2_all = _gather_rows(e1, group), _gather_rows(e2, group)
            q_offset = rank * e1.shape[0]
        else:
            e1_all, e2_all, q_offset = e1, e2, 0
        lse_row, lse_col, loss = kernels.forward(e1, e2, e1_all, e2_all, q_offset, log_scale, bias)
        if sharded:
            lse_row, lse_col = _gather_rows(lse_row, group), _gather_rows(lse_col, group)
            loss = loss.clone()
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=group)
        ctx.kernels, ctx.q_offset = kernels, q_offset
        ctx.save_for_backward(e1, e2, e1_all, e2_all, log_scale, bias, lse_row, lse_col)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        e1, e2, e1_all, e2_all, log_scale, bias, lse_row, lse_col = ctx.saved_tensors
        g = grad_out.to(torch.float32).reshape(()).contiguous()
        d1, d2, dscale, dbias = ctx.kernels.backward(e1, e2, e1_all, e2_all, ctx.q_offset, log_scale, bias,
                                                     lse_row, lse_col, g)
        return d1, d2, dscale, dbias, None, None, None


def _is_sharded(global_negatives, group):
    return bool(global_negatives and dist.is_available() and dist.is_initialized()
                and dist.get_world_size(group) > 1)


def clip_loss(embs1, embs2, logit_scale=1.0, logit_bias=0.0, image_encoder=None, lightcurve_encoder=None,
              *, global_negatives=True, group=None, kernels=HipPairKernels):
    """Symmetric softmax InfoNCE of one modality pair (ref src/loss.py:14-38; the two encoder
    keyword arguments are accepted and ignored exactly as in the reference, :19-20)."""
    dev = embs1.device
    logit_scale = torch.as_tensor(logit_scale, dtype=torch.float32, device=dev)
    logit_bias = torch.as_tensor(logit_bias, dtype=torch.float32, device=dev)
    return _PairLoss.apply(embs1, embs2, logit_scale, logit_bias, kernels, group,
                           _is_sharded(global_negatives, group))


def clip_loss_multimodal(embeddings, logit_scales=1.0, logit_biases=0.0, *, global_negatives=True, group=None,
                         kernels=HipPairKernels):
    """Sum over modality pairs i < j (ref src/loss.py:41-65); a 0-dim scale / bias is shared by
    every pair (:49-52), a vector supplies one value per pair in (0,1),(0,2),(1,2)... order."""
    m = len(embeddings)
    n_pairs = m * (m - 1) // 2
    dev = embeddings[0].device
    scales = torch.as_tensor(logit_scales, dtype=torch.float32, device=dev)
    biases = torch.as_tensor(logit_biases, dtype=torch.float32, device=dev)
    total = 0
    for k, (i, j) in enumerate(_pairs(range(m), 2)):
        s = scales if scales.dim() == 0 else scales[k]
        b = biases if biases.dim() == 0 else biases[k]
        total = total + clip_loss(embeddings[i], embeddings[j], s, b, global_negatives=global_negatives,
                                  group=group, kernels=kernels)
    if n_pairs == 0:
        raise ValueError("clip_loss_multimodal needs at least two modalities")
    return total


# ------------------------------------------------------------------------------------- sigmoid loss
class _SigmoidPair(torch.autograd.Function):
    """sigmoid_loss of one modality pair (ref src/loss.py:68-83) on msn_sigmoid_loss_fwd / _bwd."""

    @staticmethod
    def forward(ctx, e1, e2, log_scale, bias, group, sharded):
        _lib.require_gpu()
        e1, e2 = e1.contiguous(), e2.contiguous()
        if e1.shape != e2.shape:
            raise ValueError("sigmoid_loss needs the same number of rows in both modalities (labels are bs x bs)")
        log_scale = log_scale.detach().to(torch.float32).reshape(()).contiguous()
        bias = bias.detach().to(torch.float32).reshape(()).contiguous()
        if sharded:
            e1_all, e2_all = _gather_rows(e1, group), _gather_rows(e2, group)
            q_offset = dist.get_rank(group) * e1.shape[0]
        else:
            e1_all, e2_all, q_offset = e1, e2, 0
        b, D = e1.shape
        n = e1_all.shape[0]
        L = lib()
        nb = L.msn_infonce_workspace_bytes(b, b, n, n, D)
        ws = torch.empty(max(nb, 16) // 4 + 1, dtype=torch.float32, device=e1.device)
        loss = torch.empty((), dtype=torch.float32, device=e1.device)
        check(L.msn_sigmoid_loss_fwd(ptr(e1), e1.stride(0), ptr(e2), e2.stride(0), b, ptr(e1_all), e1_all.stride(0),
                                     ptr(e2_all), e2_all.stride(0), n, D, q_offset, ptr(log_scale), ptr(bias),
                                     ptr(loss), ptr(ws), nb, stream_ptr()), "msn_sigmoid_loss_fwd")
        if sharded:
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=group)
        ctx.q_offset = q_offset
        ctx.save_for_backward(e1, e2, e1_all, e2_all, log_scale, bias)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        e1, e2, e1_all, e2_all, log_scale, bias = ctx.saved_tensors
        g = grad_out.to(torch.float32).reshape(()).contiguous()
        b, D = e1.shape
        n = e1_all.shape[0]
        L = lib()
        nb = L.msn_infonce_workspace_bytes(b, b, n, n, D)
        ws = torch.empty(max(nb, 16) // 4 + 1, dtype=torch.float32, device=e1.device)
        d1, d2 = torch.empty_like(e1), torch.empty_like(e2)
        dsb = torch.empty(2, dtype=torch.float32, device=e1.device)
        check(L.msn_sigmoid_loss_bwd(ptr(e1), e1.stride(0), ptr(e2), e2.stride(0), b, ptr(e1_all), e1_all.stride(0),
                                     ptr(e2_all), e2_all.stride(0), n, D, ctx.q_offset, ptr(log_scale), ptr(bias),
                                     ptr(g), ptr(d1), D, ptr(d2), D, ptr(dsb), ptr(ws), nb, stream_ptr()),
              "msn_sigmoid_loss_bwd")
        return d1, d2, dsb[0], dsb[1], None, None


def sigmoid_loss(embs1, embs2, logit_scale=1.0, logit_bias=2.73, *, global_negatives=True, group=None):
    """Sigmoid-based CLIP loss with the reference's sign convention (ref src/loss.py:68-83); returns a
    float32 0-dim tensor (the reference returns float64: the fp64 evaluation happens inside the kernel)."""
    dev = embs1.device
    logit_scale = torch.as_tensor(logit_scale, dtype=torch.float32, device=dev)
    logit_bias = torch.as_tensor(logit_bias, dtype=torch.float32, device=dev)
    return _SigmoidPair.apply(embs1, embs2, logit_scale, logit_bias, group, _is_sharded(global_negatives, group))


def sigmoid_loss_multimodal(embeds, logit_scales=1.0, logit_biases=2.73, *, global_negatives=True, group=None):
    """Pairwise sum of sigmoid_loss (ref src/loss.py:86-107)."""
    m = len(embeds)
    if m < 2:
        raise ValueError("sigmoid_loss_multimodal needs at least two modalities")
    dev = embeds[0].device
    scales = torch.as_tensor(logit_scales, dtype=torch.float32, device=dev)
    biases = torch.as_tensor(logit_biases, dtype=torch.float32, device=dev)
    total = 0
    for k, (i, j) in enumerate(_pairs(range(m), 2)):
        s = scales if scales.dim() == 0 else scales[k]
        b = biases if biases.dim() == 0 else biases[k]
        total = total + sigmoid_loss(embeds[i], embeds[j], s, b, global_negatives=global_negatives, group=group)
    return total

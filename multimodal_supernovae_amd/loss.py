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
from . import distributed as D
from ._lib import check, lib, ptr, stream_ptr


def _device_f32(*named):
    """Every pointer handed to the C-ABI must be float32 device memory with unit column stride: a host pointer would
    fault on the GPU (there is no CPU path to fall back to), so it is refused here."""
    for name, t in named:
        if t.device.type != "cuda":
            raise _lib.MsnHipError(f"{name} must live on the GPU (got {t.device}); the contrastive loss has no CPU path")
        if t.dtype != torch.float32:
            raise _lib.MsnHipError(f"{name} must be float32 (got {t.dtype})")
        if t.dim() == 2 and t.stride(1) != 1:
            raise _lib.MsnHipError(f"{name} must have contiguous columns (strides {t.stride()})")


class HipPairKernels:
    """The product compute backend: msn_infonce_fwd / msn_infonce_bwd through the C-ABI."""

    @staticmethod
    def forward(e1_loc, e2_loc, e1_all, e2_all, q_offset, log_scale, bias):
        _lib.require_gpu()
        _device_f32(("embs1", e1_loc), ("embs2", e2_loc), ("embs1 (gathered)", e1_all), ("embs2 (gathered)", e2_all),
                    ("logit_scale", log_scale), ("logit_bias", bias))
        if e1_loc.shape[1] != e2_loc.shape[1] or e1_all.shape[1] != e1_loc.shape[1] or e2_all.shape[1] != e1_loc.shape[1]:
            raise _lib.MsnHipError("both modalities must share the embedding width")
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
        _device_f32(("embs1", e1_loc), ("embs2", e2_loc), ("embs1 (gathered)", e1_all), ("embs2 (gathered)", e2_all),
                    ("logit_scale", log_scale), ("logit_bias", bias), ("lse_row", lse_row_all), ("lse_col", lse_col_all),
                    ("grad_out", grad_out))
        if lse_row_all.numel() < e2_all.shape[0] or lse_col_all.numel() < e1_all.shape[0]:
            raise _lib.MsnHipError("backward needs the log-sum-exp of every gathered row")
        lse_row_all, lse_col_all = lse_row_all.contiguous(), lse_col_all.contiguous()
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
    return D.all_gather_rows(t, group)


class _AllReduceSum(torch.autograd.Function):
    """Loss shares of the ranks summed into the global loss; the gradient of the global loss w.r.t. a rank's share is
    the incoming gradient itself (every rank differentiates the same global scalar)."""

    @staticmethod
    def forward(ctx, x, group):
        return D.all_reduce_sum(x.clone(), group, kind="loss_all_reduce")

    @staticmethod
    def backward(ctx, g):
        return g, None


def gather_embeddings(embs, group):
    """ONE all-gather for all modalities: the local (b, D) blocks are packed side by side into a (b, M*D) send buffer,
    gathered into (world * b, M*D), and handed back as M row-strided column views."""
    b = embs[0].shape[0]
    if any(e.shape[0] != b for e in embs):
        raise ValueError("global-negatives mode needs the same local batch for every modality")
    widths = [e.shape[1] for e in embs]
    pack = torch.cat([e.detach() for e in embs], dim=1) if len(embs) > 1 else embs[0].detach()
    everything = D.all_gather_rows(pack, group, kind="embedding_all_gather")
    out, off = [], 0
    for w in widths:
        out.append(everything[:, off:off + w])
        off += w
    return out


class _MultiPairLoss(torch.autograd.Function):
    """clip_loss_multimodal as ONE node: every modality pair i < j of one step, single process or row-sharded over
    `group`.  Sharded, the step costs three collectives whatever the number of pairs -- one all-gather of the packed
    (b, M*D) embeddings, one of the packed (2P, b) row / column log-sum-exps, one all-reduce of the scalar loss --
    and each rank evaluates only its own rows and columns of every pair's logit matrix."""

    @staticmethod
    def forward(ctx, scales, biases, kernels, group, sharded, *embs):
        m = len(embs)
        pairs = list(_pairs(range(m), 2))
        embs = [e.contiguous() for e in embs]
        scales = scales.detach().to(torch.float32)
        biases = biases.detach().to(torch.float32)
        if scales.dim() > 0 and scales.numel() < len(pairs) or biases.dim() > 0 and biases.numel() < len(pairs):
            raise ValueError(f"{len(pairs)} modality pairs need {len(pairs)} logit scales / biases (or one shared scalar)")
        s_k = [(scales if scales.dim() == 0 else scales[k]).reshape(()).contiguous() for k in range(len(pairs))]
        b_k = [(biases if biases.dim() == 0 else biases[k]).reshape(()).contiguous() for k in range(len(pairs))]
        if sharded:
            alls = gather_embeddings(embs, group)
            q_offset = dist.get_rank(group) * embs[0].shape[0]
        else:
            alls, q_offset = embs, 0
        lses, losses = [], []
        for k, (i, j) in enumerate(pairs):
            lse_row, lse_col, loss = kernels.forward(embs[i], embs[j], alls[i], alls[j], q_offset, s_k[k], b_k[k])
            lses += [lse_row, lse_col]
            losses.append(loss)
        total = losses[0] if len(losses) == 1 else torch.stack(losses).sum()
        if sharded:
            b = embs[0].shape[0]
            world = dist.get_world_size(group)
            pack = torch.stack(lses)                                                        # (2P, b)
            g = D.all_gather_rows(pack.view(1, 2 * len(pairs), b), group, kind="lse_all_gather")   # (world, 2P, b)
            lse_all = g.permute(1, 0, 2).reshape(2 * len(pairs), world * b)                 # row k: rank-ordered rows
            lses = [lse_all[r] for r in range(2 * len(pairs))]
            total = D.all_reduce_sum(total.clone(), group, kind="loss_all_reduce")
        ctx.kernels, ctx.q_offset, ctx.pairs, ctx.m = kernels, q_offset, pairs, m
        ctx.shapes = (scales.shape, biases.shape)
        ctx.save_for_backward(*embs, *alls, *s_k, *b_k, *lses)
        return total

    @staticmethod
    def backward(ctx, grad_out):
        m, pairs = ctx.m, ctx.pairs
        P = len(pairs)
        t = ctx.saved_tensors
        embs, alls = t[:m], t[m:2 * m]
        s_k, b_k, lses = t[2 * m:2 * m + P], t[2 * m + P:2 * m + 2 * P], t[2 * m + 2 * P:]
        g = grad_out.to(torch.float32).reshape(()).contiguous()
        d_emb = [None] * m
        d_s, d_b = [], []
        for k, (i, j) in enumerate(pairs):
            d1, d2, ds, db = ctx.kernels.backward(embs[i], embs[j], alls[i], alls[j], ctx.q_offset, s_k[k], b_k[k],
                                                  lses[2 * k], lses[2 * k + 1], g)
            d_emb[i] = d1 if d_emb[i] is None else d_emb[i].add_(d1)
            d_emb[j] = d2 if d_emb[j] is None else d_emb[j].add_(d2)
            d_s.append(ds)
            d_b.append(db)
        s_shape, b_shape = ctx.shapes
        if P == 1:
            gs, gb = d_s[0].reshape(s_shape), d_b[0].reshape(b_shape)
        else:
            gs, gb = torch.stack(d_s), torch.stack(d_b)
            gs = gs.sum() if len(s_shape) == 0 else gs
            gb = gb.sum() if len(b_shape) == 0 else gb
        return (gs, gb, None, None, None, *d_emb)


class _PairLoss(torch.autograd.Function):
    """One modality pair; single process or row-sharded over `group`."""

    @staticmethod
    def forward(ctx, e1, e2, log_scale, bias, kernels, group, sharded):
        e1 = e1.contiguous()
        e2 = e2.contiguous()
        log_scale = log_scale.detach().to(torch.float32).reshape(()).contiguous()
        bias = bias.detach().to(torch.float32).reshape(()).contiguous()
        if sharded:
            rank = dist.get_rank(group)
            e1_all, e2_all = gather_embeddings([e1, e2], group)
            q_offset = rank * e1.shape[0]
        else:
            e1_all, e2_all, q_offset = e1, e2, 0
        lse_row, lse_col, loss = kernels.forward(e1, e2, e1_all, e2_all, q_offset, log_scale, bias)
        if sharded:
            both = _gather_rows(torch.stack([lse_row, lse_col]).view(1, 2, -1), group)      # (world, 2, b)
            both = both.permute(1, 0, 2).reshape(2, -1)
            lse_row, lse_col = both[0], both[1]
            loss = D.all_reduce_sum(loss.clone(), group, kind="loss_all_reduce")
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
    if not (global_negatives and dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or global_negatives == "always"     # "always": also a one-rank group (tests)


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
    if m < 2:
        raise ValueError("clip_loss_multimodal needs at least two modalities")
    dev = embeddings[0].device
    scales = torch.as_tensor(logit_scales, dtype=torch.float32, device=dev)
    biases = torch.as_tensor(logit_biases, dtype=torch.float32, device=dev)
    return _MultiPairLoss.apply(scales, biases, kernels, group, _is_sharded(global_negatives, group), *embeddings)


# ------------------------------------------------------------------------------------- sigmoid loss
class _SigmoidPair(torch.autograd.Function):
    """sigmoid_loss of one modality pair (ref src/loss.py:68-83) on msn_sigmoid_loss_fwd / _bwd."""

    @staticmethod
    def forward(ctx, e1, e2, log_scale, bias, group, sharded, gathered):
        _lib.require_gpu()
        e1, e2 = e1.contiguous(), e2.contiguous()
        if e1.shape != e2.shape:
            raise ValueError("sigmoid_loss needs the same number of rows in both modalities (labels are bs x bs)")
        log_scale = log_scale.detach().to(torch.float32).reshape(()).contiguous()
        bias = bias.detach().to(torch.float32).reshape(()).contiguous()
        if sharded:
            # `gathered`: this pair's columns of the step's ONE packed all-gather (sigmoid_loss_multimodal)
            e1_all, e2_all = gathered if gathered is not None else gather_embeddings([e1, e2], group)
            q_offset = dist.get_rank(group) * e1.shape[0]
        else:
            e1_all, e2_all, q_offset = e1, e2, 0
        _device_f32(("embs1", e1), ("embs2", e2), ("embs1 (gathered)", e1_all), ("embs2 (gathered)", e2_all),
                    ("logit_scale", log_scale), ("logit_bias", bias))
        b, D = e1.shape
        n = e1_all.shape[0]
        L = lib()
        nb = L.msn_infonce_workspace_bytes(b, b, n, n, D)
        ws = torch.empty(max(nb, 16) // 4 + 1, dtype=torch.float32, device=e1.device)
        loss = torch.empty((), dtype=torch.float32, device=e1.device)
        check(L.msn_sigmoid_loss_fwd(ptr(e1), e1.stride(0), ptr(e2), e2.stride(0), b, ptr(e1_all), e1_all.stride(0),
                                     ptr(e2_all), e2_all.stride(0), n, D, q_offset, ptr(log_scale), ptr(bias),
                                     ptr(loss), ptr(ws), nb, stream_ptr()), "msn_sigmoid_loss_fwd")
        ctx.q_offset = q_offset                      # the rank's SHARE of the loss: the caller sums the ranks once
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
        return d1, d2, dsb[0], dsb[1], None, None, None


def sigmoid_loss(embs1, embs2, logit_scale=1.0, logit_bias=2.73, *, global_negatives=True, group=None):
    """Sigmoid-based CLIP loss with the reference's sign convention (ref src/loss.py:68-83); returns a
    float32 0-dim tensor (the reference returns float64: the fp64 evaluation happens inside the kernel)."""
    dev = embs1.device
    logit_scale = torch.as_tensor(logit_scale, dtype=torch.float32, device=dev)
    logit_bias = torch.as_tensor(logit_bias, dtype=torch.float32, device=dev)
    sharded = _is_sharded(global_negatives, group)
    share = _SigmoidPair.apply(embs1, embs2, logit_scale, logit_bias, group, sharded, None)
    return _AllReduceSum.apply(share, group) if sharded else share


def sigmoid_loss_multimodal(embeds, logit_scales=1.0, logit_biases=2.73, *, global_negatives=True, group=None):
    """Pairwise sum of sigmoid_loss (ref src/loss.py:86-107)."""
    m = len(embeds)
    if m < 2:
        raise ValueError("sigmoid_loss_multimodal needs at least two modalities")
    dev = embeds[0].device
    scales = torch.as_tensor(logit_scales, dtype=torch.float32, device=dev)
    biases = torch.as_tensor(logit_biases, dtype=torch.float32, device=dev)
    sharded = _is_sharded(global_negatives, group)
    alls = gather_embeddings([e.contiguous() for e in embeds], group) if sharded else None
    total = 0
    for k, (i, j) in enumerate(_pairs(range(m), 2)):
        s = scales if scales.dim() == 0 else scales[k]
        b = biases if biases.dim() == 0 else biases[k]
        total = total + _SigmoidPair.apply(embeds[i], embeds[j], s, b, group, sharded,
                                           (alls[i], alls[j]) if sharded else None)
    return _AllReduceSum.apply(total, group) if sharded else total

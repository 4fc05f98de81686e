"""Differentiable building blocks: torch.autograd.Function nodes whose forward AND backward are
sequences of libmsn_hip kernels (ops.py).  One node per transformer block / linear chain / pooling
step, so residual adds, bias adds, activations and their derivatives ride in GEMM epilogues and no
stock ATen arithmetic runs on the hot path.  Reference lines are cited per node."""
import math

import torch

from . import ops
from .ops import (EPI_ADD, EPI_GELU, EPI_GELU_BWD, EPI_NONE, EPI_RELU, EPI_RELU_BWD, OP_N, OP_T, colsum, sgemm)

ACT_NONE, ACT_RELU, ACT_GELU = "none", "relu", "gelu"


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _remember_precision(cls):
    """Class decorator: the GEMM precision in force when a node's forward ran is re-established for its
    backward (which autograd calls later, outside any ops.gemm_precision(...) scope)."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx._gemm_precision = ops.GEMM_PRECISION
        return fwd(ctx, *args)

    def backward(ctx, *grads):
        with ops.gemm_precision(ctx._gemm_precision):
            return bwd(ctx, *grads)

    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


# ------------------------------------------------------------------------------------- dropout
class _Dropout(torch.autograd.Function):
    """nn.Dropout in train mode; the mask is a function of (seed, element index) and is recomputed for the
    gradient instead of being stored."""

    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.cfg = (p, seed)
        return ops.dropout(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.cfg
        return ops.dropout(dy, p, seed), None, None


def dropout(x, p, training=True):
    if not training or p <= 0.0:
        return x
    return _Dropout.apply(x, float(p), ops.new_seed())


# ------------------------------------------------------------------------------- linear chains
@_remember_precision
class _LinearChain(torch.autograd.Function):
    """y = L_n(act_{n-1}(... act_1(L_1(x)))) -- nn.Linear stacks with ReLU / GELU between them
    (MLP, ref models_multimodal.py:853-856; ConvMixer head :85-88; any single Linear).  The
    activation derivative of layer i is applied in the epilogue of layer i+1's dgrad GEMM."""

    @staticmethod
    def forward(ctx, x, acts, drop_p, *wb):
        n = len(acts)
        shape = x.shape
        h = _c(x).view(-1, shape[-1])
        inputs, pres, seeds = [], [], []
        for i in range(n):
            w, b = wb[2 * i], wb[2 * i + 1]
            inputs.append(h)
            if acts[i] == ACT_GELU:
                pre = torch.empty((h.shape[0], w.shape[0]), dtype=torch.float32, device=h.device)
                h = sgemm(h, w, OP_N, OP_T, bias=b, epilogue=EPI_GELU, aux=pre)
                pres.append(pre)
            elif acts[i] == ACT_RELU:
                h = sgemm(h, w, OP_N, OP_T, bias=b, epilogue=EPI_RELU)
                pres.append(None)
            else:
                h = sgemm(h, w, OP_N, OP_T, bias=b)
                pres.append(None)
            seeds.append(None)
            if drop_p > 0.0 and acts[i] != ACT_NONE:       # Linear -> act -> Dropout (ref :850, :87)
                seeds[i] = ops.new_seed()
                ops.dropout(h, drop_p, seeds[i], out=h)
        if acts[-1] != ACT_NONE:
            raise ValueError("the last layer of a linear chain carries no activation")
        ctx.acts, ctx.shape = acts, shape
        ctx.drop = (drop_p, seeds)
        ctx.n_saved_in = n
        ctx.save_for_backward(*inputs, *[p if p is not None else torch.empty(0) for p in pres], *wb)
        return h.view(*shape[:-1], h.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        n = ctx.n_saved_in
        saved = ctx.saved_tensors
        inputs, pres, wb = saved[:n], saved[n:2 * n], saved[2 * n:]
        d = _c(dy).view(-1, dy.shape[-1])
        grads = [None] * (2 * n)
        for i in range(n - 1, -1, -1):
            w, b = wb[2 * i], wb[2 * i + 1]
            if b is not None:
                grads[2 * i], grads[2 * i + 1] = ops.wgrad_bias(d, inputs[i])     # dW = d^T . input, db = sum d
            else:
                grads[2 * i] = sgemm(d, inputs[i], OP_T, OP_N)
            if i > 0:
                act = ctx.acts[i - 1]
                if act == ACT_RELU:      # inputs[i] = relu output of layer i-1 (zero where dropped: same gate)
                    d = sgemm(d, w, OP_N, OP_N, epilogue=EPI_RELU_BWD, aux=inputs[i])
                elif act == ACT_GELU:
                    d = sgemm(d, w, OP_N, OP_N, epilogue=EPI_GELU_BWD, aux=pres[i - 1])
                else:
                    d = sgemm(d, w, OP_N, OP_N)
                drop_p, seeds = ctx.drop
                if seeds[i - 1] is not None:     # gradient through the dropout that followed the activation
                    ops.dropout(d, drop_p, seeds[i - 1], out=d)
            elif ctx.needs_input_grad[0]:
                d = sgemm(d, w, OP_N, OP_N)
            else:
                d = None
        dx = d.view(ctx.shape) if d is not None else None
        return (dx, None, None, *grads)


def linear_chain(x, layers, acts, drop_p=0.0):
    """layers: [(weight, bias_or_None), ...]; acts: one of none/relu/gelu per layer (last = none);
    drop_p > 0 applies dropout after every activation (train mode)."""
    flat = []
    for w, b in layers:
        flat += [w, b]
    return _LinearChain.apply(x, tuple(acts), float(drop_p), *flat)


def linear(x, weight, bias=None):
    return _LinearChain.apply(x, (ACT_NONE,), 0.0, weight, bias)


# --------------------------------------------------------------------- projection + L2 normalise
@_remember_precision
class _ProjectNormalise(torch.autograd.Function):
    """x -> Linear(n_out, enc_dim) -> x / ||x||  (ref models_multimodal.py:275-304, no epsilon)."""

    @staticmethod
    def forward(ctx, h, w, b):
        h = _c(h)
        z = sgemm(h, w, OP_N, OP_T, bias=b)
        y, inv = ops.l2norm_fwd(z)
        ctx.save_for_backward(h, w, y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, w, y, inv = ctx.saved_tensors
        dz = ops.l2norm_bwd(_c(dy), y, inv)
        return sgemm(dz, w, OP_N, OP_N), sgemm(dz, h, OP_T, OP_N), colsum(dz)


def project_normalise(h, weight, bias):
    return _ProjectNormalise.apply(h, weight, bias)


class _L2Normalise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, inv = ops.l2norm_fwd(_c(x))
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        return ops.l2norm_bwd(_c(dy), y, inv)


def l2_normalise(x):
    return _L2Normalise.apply(x)


# ------------------------------------------------------------------------------ self-attention
def _attn_forward_raw(x2, B, T, wq, wk, wv, wu, bu, mask_u8, heads, scale, residual, wcat=None):
    """x2: (B*T, e).  Returns z = unify(attn) (+ residual) and what backward needs."""
    M, e = x2.shape
    # the three bias-free projections read the same tokens: one product against the stacked (3e, e) weight instead of
    # three passes over x2; `wcat` is the module's persistent stacked buffer (SelfAttention.stacked_qkv: the three
    # parameters ARE its row blocks), a copy is made only for callers that hand over loose weights
    if wcat is None:
        wcat = torch.cat([wq, wk, wv], 0)
    qkv = sgemm(x2, wcat, OP_N, OP_T)
    q3 = qkv.view(B, T, 3 * e)
    a, lse = ops.attention_fwd(q3[..., :e], q3[..., e:2 * e], q3[..., 2 * e:], mask_u8, heads, scale)
    a2 = a.view(M, e)
    if residual is not None:
        z = sgemm(a2, wu, OP_N, OP_T, bias=bu, epilogue=EPI_ADD, aux=residual)
    else:
        z = sgemm(a2, wu, OP_N, OP_T, bias=bu)
    return z, (qkv, a2, lse, wcat)


def _attn_backward_raw(dz, x2, B, T, saved, wq, wk, wv, wu, mask_u8, heads, scale, add_to):
    """dz: grad of unify output.  Returns dx (+ add_to fused) and the parameter grads."""
    qkv, a2, lse, wcat = saved
    M, e = x2.shape
    dwu, dbu = ops.wgrad_bias(dz, a2)
    da = sgemm(dz, wu, OP_N, OP_N)
    dqkv = torch.empty_like(qkv)
    q3, d3 = qkv.view(B, T, 3 * e), dqkv.view(B, T, 3 * e)
    ops.attention_bwd(q3[..., :e], q3[..., e:2 * e], q3[..., 2 * e:], mask_u8, heads, scale, a2.view(B, T, e), lse,
                      da.view(B, T, e), d3[..., :e], d3[..., e:2 * e], d3[..., 2 * e:])
    dw = sgemm(dqkv, x2, OP_T, OP_N)                       # stacked (3e, e): rows = dWq | dWk | dWv
    if add_to is not None:
        dx = sgemm(dqkv, wcat, OP_N, OP_N, epilogue=EPI_ADD, aux=add_to)
    else:
        dx = sgemm(dqkv, wcat, OP_N, OP_N)
    return dx, dw[:e], dw[e:2 * e], dw[2 * e:], dwu, dbu


@_remember_precision
class _SelfAttention(torch.autograd.Function):
    """ref transformer_utils.py:36-89 as one node (q/k/v GEMMs, fused attention, unifyheads)."""

    @staticmethod
    def forward(ctx, x, mask_u8, heads, wk, wq, wv, wu, bu, wcat):
        B, T, e = x.shape
        x2 = _c(x).view(B * T, e)
        scale = 1.0 / math.sqrt(e)
        z, saved = _attn_forward_raw(x2, B, T, wq, wk, wv, wu, bu, mask_u8, heads, scale, None, wcat)
        ctx.dims = (B, T, e, heads, scale)
        ctx.mask = mask_u8
        ctx.save_for_backward(x2, wq, wk, wv, wu, *saved)
        return z.view(B, T, e)

    @staticmethod
    def backward(ctx, dy):
        B, T, e, heads, scale = ctx.dims
        x2, wq, wk, wv, wu, qkv, a2, lse, wcat = ctx.saved_tensors
        dz = _c(dy).view(B * T, e)
        dx, dwq, dwk, dwv, dwu, dbu = _attn_backward_raw(dz, x2, B, T, (qkv, a2, lse, wcat), wq, wk, wv, wu, ctx.mask,
                                                         heads, scale, None)
        return dx.view(B, T, e), None, None, dwk, dwq, dwv, dwu, dbu, None


def self_attention(x, mask, heads, tokeys, toqueries, tovalues, unify_w, unify_b, wcat=None):
    return _SelfAttention.apply(x, ops._mask_u8(mask), heads, tokeys, toqueries, tovalues, unify_w, unify_b, wcat)


# ---------------------------------------------------------------- post-norm transformer block
FUSED_FF = True       # False (tests, A/B runs): the feed-forward half as four msn_sgemm products with the hidden matrix in memory


@_remember_precision
class _PostNormBlock(torch.autograd.Function):
    """ref transformer_utils.py:109-116: x = LN1(attn(x) + x); x = LN2(FF(x) + x), FF = Linear ->
    ReLU -> Linear.  Residual adds ride in the unifyheads / ff.2 GEMM epilogues; in backward the
    residual-branch gradients ride in the dgrad epilogues."""

    @staticmethod
    def forward(ctx, x, mask_u8, heads, drop_p, wk, wq, wv, wu, bu, g1, b1, w1, c1, w2, c2, g2, b2, wcat, ff_planes=None):
        B, T, e = x.shape
        x2 = _c(x).view(B * T, e)
        scale = 1.0 / math.sqrt(e)
        z1, (qkv, a2, lse, wcat) = _attn_forward_raw(x2, B, T, wq, wk, wv, wu, bu, mask_u8, heads, scale, x2, wcat)
        y1, m1, r1 = ops.layernorm_fwd(z1, g1, b1)
        seeds = None
        if drop_p > 0.0:                 # x = do(norm1(...)) / x = do(norm2(...)), ref :111-116
            seeds = (ops.new_seed(), ops.new_seed())
            ops.dropout(y1, drop_p, seeds[0], out=y1)
        # the feed-forward half: one kernel per direction whose 4e-wide hidden matrix stays on chip where the fused kernels take the
        # width (emb 32: the spectrum transformer) and the process arithmetic is the plane one they compute in (csrc/ffn_planes.hip)
        fused_ff = FUSED_FF and ops.plane_count() == 3 and ops.ffn_supported(B * T, e, w1.shape[0])
        if fused_ff:
            # (ff_planes: the two plane matrices handed in by the tower, which splits the weights of ALL its blocks in one launch)
            w1p, w2tp = ff_planes if ff_planes is not None else ops.ffn_weight_planes(w1, w2)
            z2 = ops.ffn_fwd(y1, w1p, w2tp, c1, c2)
            hdn = y1.new_empty(0)
            ctx.ff_planes = (w1p, w2tp)
        else:
            hdn = sgemm(y1, w1, OP_N, OP_T, bias=c1, epilogue=EPI_RELU)
            z2 = sgemm(hdn, w2, OP_N, OP_T, bias=c2, epilogue=EPI_ADD, aux=y1)
            ctx.ff_planes = None
        y2, m2, r2 = ops.layernorm_fwd(z2, g2, b2)
        if seeds:
            ops.dropout(y2, drop_p, seeds[1], out=y2)
        ctx.drop = (drop_p, seeds)
        ctx.dims = (B, T, e, heads, scale)
        ctx.mask = mask_u8
        ctx.save_for_backward(x2, wq, wk, wv, wu, g1, w1, w2, g2, qkv, a2, lse, z1, m1, r1, y1, hdn, z2, m2, r2, wcat, c1)
        return y2.view(B, T, e)

    @staticmethod
    def backward(ctx, dy):
        B, T, e, heads, scale = ctx.dims
        (x2, wq, wk, wv, wu, g1, w1, w2, g2, qkv, a2, lse, z1, m1, r1, y1, hdn, z2, m2, r2, wcat, c1) = ctx.saved_tensors
        dy2 = _c(dy).view(B * T, e)
        drop_p, seeds = ctx.drop
        if seeds:
            dy2 = ops.dropout(dy2, drop_p, seeds[1])
        dz2, dg2, db2 = ops.layernorm_bwd(dy2, z2, m2, r2, g2)
        if ctx.ff_planes is not None:     # hidden tile recomputed on chip; dy1 comes with the residual branch of LN2's input added
            dy1, dw1, dc1, dw2, dc2 = ops.ffn_bwd(y1, dz2, ctx.ff_planes[0], ctx.ff_planes[1], c1)
        else:
            dw2, dc2 = ops.wgrad_bias(dz2, hdn)
            dpre = sgemm(dz2, w2, OP_N, OP_N, epilogue=EPI_RELU_BWD, aux=hdn)
            dw1, dc1 = ops.wgrad_bias(dpre, y1)
            dy1 = sgemm(dpre, w1, OP_N, OP_N, epilogue=EPI_ADD, aux=dz2)   # + residual branch of LN2's input
        if seeds:
            ops.dropout(dy1, drop_p, seeds[0], out=dy1)
        dz1, dg1, db1 = ops.layernorm_bwd(dy1, z1, m1, r1, g1)
        dx, dwq, dwk, dwv, dwu, dbu = _attn_backward_raw(dz1, x2, B, T, (qkv, a2, lse, wcat), wq, wk, wv, wu, ctx.mask,
                                                         heads, scale, dz1)  # + residual branch of LN1's input
        return (dx.view(B, T, e), None, None, None, dwk, dwq, dwv, dwu, dbu, dg1, db1, dw1, dc1, dw2, dc2, dg2, db2, None, None)


def fused_ff_applies(x, emb, hidden):
    """Will post_norm_block run its feed-forward half as the fused kernels for a (B, T, emb) input?"""
    return FUSED_FF and x.is_cuda and ops.plane_count() == 3 and ops.ffn_supported(x.shape[0] * x.shape[1], emb, hidden)


def post_norm_block(x, mask_u8, heads, p, drop_p=0.0, wcat=None, ff_planes=None):
    """p: dict with tokeys, toqueries, tovalues, unify_w, unify_b, norm1_w, norm1_b, ff0_w, ff0_b,
    ff2_w, ff2_b, norm2_w, norm2_b.  drop_p > 0: dropout after both LayerNorms (train mode).
    wcat: the persistent stacked [toqueries ; tokeys ; tovalues] matrix (SelfAttention.stacked_qkv) or None.
    ff_planes: (planes of ff0_w, planes of ff2_w transposed) when the caller has split them already (ops.ffn_weight_planes)."""
    return _PostNormBlock.apply(x, mask_u8, heads, float(drop_p), p["tokeys"], p["toqueries"], p["tovalues"], p["unify_w"],
                                p["unify_b"], p["norm1_w"], p["norm1_b"], p["ff0_w"], p["ff0_b"], p["ff2_w"],
                                p["ff2_b"], p["norm2_w"], p["norm2_b"], wcat, ff_planes)


# ------------------------------------------------------------------- time / band embedding
class _TimeEmbed(torch.autograd.Function):
    """ref transformer_utils.py:214-231: Linear(1, emb)(x) + sin/cos(t * omega) (+ band embedding)."""

    @staticmethod
    def forward(ctx, x, t, w, bw, band, omega):
        x, t = _c(x), _c(t)
        out = ops.time_embed_fwd(x, t, _c(w).view(-1), bw, omega, band)
        ctx.nband = band.shape[0] if band is not None else 1
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dw, dbw, dband = ops.time_embed_bwd(_c(dy), x, ctx.nband)
        return None, None, dw.view(-1, 1), dbw, dband, None


def time_embed(x, t, mag_w, mag_b, band_w, omega):
    return _TimeEmbed.apply(x, t, mag_w, mag_b, band_w, omega)


# ----------------------------------------------------------------------------------- pooling
class _MaskedPool(torch.autograd.Function):
    """ref transformer_utils.py:234-239 (zero padded tokens, then mean over valid / max over all)."""

    @staticmethod
    def forward(ctx, x, mask_u8, mode):
        out, arg, cnt = ops.masked_pool_fwd(_c(x), mask_u8, mode)
        ctx.mode, ctx.T, ctx.mask = mode, x.shape[1], mask_u8
        ctx.save_for_backward(arg if arg is not None else torch.empty(0), cnt if cnt is not None else torch.empty(0))
        return out

    @staticmethod
    def backward(ctx, dout):
        arg, cnt = ctx.saved_tensors
        dx = ops.masked_pool_bwd(_c(dout), ctx.mask, ctx.T, ctx.mode, arg if ctx.mode == ops.POOL_MAX else None,
                                 cnt if ctx.mode == ops.POOL_MEAN else None)
        return dx, None, None


def masked_pool(x, mask_u8, mode):
    return _MaskedPool.apply(x, mask_u8, ops.POOL_MEAN if mode == "mean" else ops.POOL_MAX)


class _MaskTokens(torch.autograd.Function):
    """x * mask[:, :, None] (ref transformer_utils.py:235)."""

    @staticmethod
    def forward(ctx, x, mask_u8):
        ctx.mask = mask_u8
        return ops.mask_tokens(_c(x), mask_u8)

    @staticmethod
    def backward(ctx, dy):
        return ops.mask_tokens(_c(dy), ctx.mask), None


def mask_tokens(x, mask_u8):
    return _MaskTokens.apply(x, mask_u8)


@_remember_precision
class _AttnPool(torch.autograd.Function):
    """ref transformer_utils.py:240-246: nn.MultiheadAttention(emb, 2 heads, batch_first) with ONE
    learnable query shared by the batch, keys = values = the (zeroed) tokens, no key mask."""

    HEADS = 2

    @staticmethod
    def forward(ctx, x, query, w_in, b_in, w_out, b_out):
        B, T, e = x.shape
        x2 = _c(x).view(B * T, e)
        q0 = _c(query).view(1, e)
        qp = sgemm(q0, w_in[:e], OP_N, OP_T, bias=b_in[:e])                  # (1, e)
        kv = torch.empty((B * T, 2 * e), dtype=torch.float32, device=x.device)
        sgemm(x2, w_in[e:2 * e], OP_N, OP_T, bias=b_in[e:2 * e], out=kv[:, :e])
        sgemm(x2, w_in[2 * e:], OP_N, OP_T, bias=b_in[2 * e:], out=kv[:, e:])
        kv3 = kv.view(B, T, 2 * e)
        scale = 1.0 / math.sqrt(e // _AttnPool.HEADS)
        a, lse = ops.attention_fwd(qp.view(1, 1, e), kv3[..., :e], kv3[..., e:], None, _AttnPool.HEADS, scale,
                                   q_shared=True)
        a2 = a.view(B, e)
        out = sgemm(a2, w_out, OP_N, OP_T, bias=b_out)
        ctx.dims = (B, T, e, scale)
        ctx.save_for_backward(x2, q0, w_in, w_out, qp, kv, a2, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, e, scale = ctx.dims
        x2, q0, w_in, w_out, qp, kv, a2, lse = ctx.saved_tensors
        dout = _c(dout)
        dw_out, db_out = ops.wgrad_bias(dout, a2)
        da = sgemm(dout, w_out, OP_N, OP_N)
        kv3 = kv.view(B, T, 2 * e)
        dq = torch.empty((B, 1, e), dtype=torch.float32, device=dout.device)
        dkv = torch.empty_like(kv)
        d3 = dkv.view(B, T, 2 * e)
        ops.attention_bwd(qp.view(1, 1, e), kv3[..., :e], kv3[..., e:], None, _AttnPool.HEADS, scale,
                          a2.view(B, 1, e), lse, da.view(B, 1, e), dq, d3[..., :e], d3[..., e:], q_shared=True)
        dqp = colsum(dq.view(B, e)).view(1, e)
        dw_in = torch.empty_like(w_in)
        db_in = torch.empty(3 * e, dtype=torch.float32, device=dout.device)
        sgemm(dqp, q0, OP_T, OP_N, out=dw_in[:e])
        sgemm(dkv[:, :e], x2, OP_T, OP_N, out=dw_in[e:2 * e])
        sgemm(dkv[:, e:], x2, OP_T, OP_N, out=dw_in[2 * e:])
        db_in[:e].copy_(dqp.view(-1))
        db_in[e:] = colsum(dkv)
        dquery = sgemm(dqp, w_in[:e], OP_N, OP_N).view(-1)
        dx = sgemm(dkv[:, :e], w_in[e:2 * e], OP_N, OP_N)
        sgemm(dkv[:, e:], w_in[2 * e:], OP_N, OP_N, epilogue=EPI_ADD, aux=dx, out=dx)
        return dx.view(B, T, e), dquery, dw_in, db_in, dw_out, db_out


def attn_pool(x, query, in_proj_weight, in_proj_bias, out_w, out_b):
    return _AttnPool.apply(x, query, in_proj_weight, in_proj_bias, out_w, out_b)


# ------------------------------------------------------------------------------ ConvMixer trunk
@_remember_precision
class _ConvMixerTrunk(torch.autograd.Function):
    """ref models_multimodal.py:52-79 (the `net` Sequential): patch conv -> GELU -> BN, then per
    layer x + BN(GELU(depthwise(x))) followed by BN(GELU(1x1 conv)).  Channels-last token matrices;
    convs with dense weights are GEMMs (GELU in the epilogue), the residual add rides in the BN
    apply kernel, GELU' rides in the BN backward kernel.  Returns tokens (B, gh*gw, dim).

    flat parameter order: w0, g0, b0, rm0, rv0, then per layer
      dw_w, dw_b, gA, bA, rmA, rvA, pw_w, pw_b, gB, bB, rmB, rvB."""

    PER_LAYER = 12

    @staticmethod
    def forward(ctx, img, training, depth, patch, drop_p, *P):
        B, C, H, W = img.shape
        drop_p = drop_p if training else 0.0
        seeds = []
        gh, gw = H // patch, W // patch
        img = _c(img.float())
        w0, g0, b0, rm0, rv0 = P[:5]
        dim = w0.shape[0]
        patches = ops.patchify(img, patch)
        M = patches.shape[0]
        pre0 = torch.empty((M, dim), dtype=torch.float32, device=img.device)
        act0 = sgemm(patches, w0.view(dim, -1), OP_N, OP_T, epilogue=EPI_GELU, aux=pre0)
        y, mean0, rstd0 = ops.batchnorm_fwd(act0, g0, b0, rm0, rv0, training)
        saved = [patches, pre0, act0, mean0, rstd0]
        for i in range(depth):
            dw_w, dw_b, gA, bA, rmA, rvA, pw_w, pw_b, gB, bB, rmB, rvB = P[5 + 12 * i: 17 + 12 * i]
            preA, actA = ops.dwconv_gelu_fwd(y, dw_w, dw_b, B, gh, gw)
            if drop_p > 0.0:      # x + Dropout(BN(GELU(dw(x)))), then Dropout(BN(GELU(1x1))) -- ref :62-79
                sA, sB = ops.new_seed(), ops.new_seed()
                seeds.append((sA, sB))
                yA, meanA, rstdA = ops.batchnorm_fwd(actA, gA, bA, rmA, rvA, training)
                ops.dropout(yA, drop_p, sA, residual=y, out=yA)
            else:
                yA, meanA, rstdA = ops.batchnorm_fwd(actA, gA, bA, rmA, rvA, training, residual=y)
            preB = torch.empty((M, dim), dtype=torch.float32, device=img.device)
            actB = sgemm(yA, pw_w.view(dim, dim), OP_N, OP_T, bias=pw_b, epilogue=EPI_GELU, aux=preB)
            yB, meanB, rstdB = ops.batchnorm_fwd(actB, gB, bB, rmB, rvB, training)
            if drop_p > 0.0:
                ops.dropout(yB, drop_p, sB, out=yB)
            saved += [y, preA, actA, meanA, rstdA, yA, preB, actB, meanB, rstdB]
            y = yB
        ctx.geom = (B, C, H, W, gh, gw, dim, depth, patch, bool(training))
        ctx.drop = (drop_p, seeds)
        ctx.n_act = len(saved)
        ctx.save_for_backward(*saved, *P)
        return y.view(B, gh * gw, dim)

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, gh, gw, dim, depth, patch, training = ctx.geom
        saved = ctx.saved_tensors
        acts, P = saved[:ctx.n_act], saved[ctx.n_act:]
        grads = [None] * len(P)
        d = _c(dy).view(B * gh * gw, dim)
        for i in range(depth - 1, -1, -1):
            o = 5 + 12 * i
            dw_w, dw_b, gA, bA, rmA, rvA, pw_w, pw_b, gB, bB, rmB, rvB = P[o:o + 12]
            y_in, preA, actA, meanA, rstdA, yA, preB, actB, meanB, rstdB = acts[5 + 10 * i: 15 + 10 * i]
            drop_p, seeds = ctx.drop
            if drop_p > 0.0:
                d = ops.dropout(d, drop_p, seeds[i][1])
            dpreB, dgB, dbB = ops.batchnorm_bwd(d, actB, preB, meanB, rstdB, gB, training)
            dpw, grads[o + 7] = ops.wgrad_bias(dpreB, yA)
            grads[o + 6] = dpw.view_as(pw_w)
            grads[o + 8], grads[o + 9] = dgB, dbB
            dyA = sgemm(dpreB, pw_w.view(dim, dim), OP_N, OP_N)
            dbn = ops.dropout(dyA, drop_p, seeds[i][0]) if drop_p > 0.0 else dyA      # branch through the dropout
            dpreA, dgA, dbA = ops.batchnorm_bwd(dbn, actA, preA, meanA, rstdA, gA, training)
            d, ddw, ddb = ops.dwconv_bwd(dpreA, y_in, dw_w, B, gh, gw, add=dyA)   # + the residual branch
            grads[o], grads[o + 1], grads[o + 2], grads[o + 3] = ddw, ddb, dgA, dbA
        patches, pre0, act0, mean0, rstd0 = acts[:5]
        w0, g0 = P[0], P[1]
        dpre0, dg0, db0 = ops.batchnorm_bwd(d, act0, pre0, mean0, rstd0, g0, training)
        grads[0] = sgemm(dpre0, patches, OP_T, OP_N).view_as(w0)
        grads[1], grads[2] = dg0, db0
        dimg = None
        if ctx.needs_input_grad[0]:
            dimg = ops.unpatchify(sgemm(dpre0, w0.view(dim, -1), OP_N, OP_N), (B, C, H, W), patch)
        return (dimg, None, None, None, None, *grads)


def convmixer_trunk(img, training, depth, patch, flat_params, drop_p=0.0):
    return _ConvMixerTrunk.apply(img, training, depth, patch, float(drop_p), *flat_params)


# ------------------------------------------------------------- pre-norm (ViT) transformer block
@_remember_precision
class _PreNormBlock(torch.autograd.Function):
    """Build-defined ViT block (not in the reference): x += Attn(LN1(x)); x += MLP_GELU(LN2(x)),
    packed qkv projection with bias, scale 1/sqrt(head_dim), no mask.  Same fusion plan as the
    post-norm block: residual adds in GEMM epilogues / the LayerNorm backward kernel."""

    @staticmethod
    def forward(ctx, x, heads, eps, g1, b1, wqkv, bqkv, wo, bo, g2, b2, w1, c1, w2, c2):
        B, T, e = x.shape
        x2 = _c(x).view(B * T, e)
        scale = 1.0 / math.sqrt(e // heads)
        h1, m1, r1 = ops.layernorm_fwd(x2, g1, b1, eps)
        qkv = sgemm(h1, wqkv, OP_N, OP_T, bias=bqkv)
        q3 = qkv.view(B, T, 3 * e)
        a, lse = ops.attention_fwd(q3[..., :e], q3[..., e:2 * e], q3[..., 2 * e:], None, heads, scale)
        a2 = a.view(B * T, e)
        x1 = sgemm(a2, wo, OP_N, OP_T, bias=bo, epilogue=EPI_ADD, aux=x2)
        h2, m2, r2 = ops.layernorm_fwd(x1, g2, b2, eps)
        pre = torch.empty((B * T, w1.shape[0]), dtype=torch.float32, device=x.device)
        f = sgemm(h2, w1, OP_N, OP_T, bias=c1, epilogue=EPI_GELU, aux=pre)
        out = sgemm(f, w2, OP_N, OP_T, bias=c2, epilogue=EPI_ADD, aux=x1)
        ctx.dims = (B, T, e, heads, scale)
        ctx.save_for_backward(x2, g1, wqkv, wo, g2, w1, w2, m1, r1, h1, qkv, a2, lse, x1, m2, r2, h2, pre, f)
        return out.view(B, T, e)

    @staticmethod
    def backward(ctx, dy):
        B, T, e, heads, scale = ctx.dims
        (x2, g1, wqkv, wo, g2, w1, w2, m1, r1, h1, qkv, a2, lse, x1, m2, r2, h2, pre, f) = ctx.saved_tensors
        d2 = _c(dy).view(B * T, e)
        pair = _pair_backward(B * T)
        if pair & 4:    # both backward products of a Linear in one work-list launch (ops.dgrad_wgrad)
            dpre, dw2, dc2 = ops.dgrad_wgrad(d2, w2, f, epilogue=EPI_GELU_BWD, aux=pre)
        else:
            dw2, dc2 = ops.wgrad_bias(d2, f)
            dpre = sgemm(d2, w2, OP_N, OP_N, epilogue=EPI_GELU_BWD, aux=pre)
        if pair & 1:
            dh2, dw1, dc1 = ops.dgrad_wgrad(dpre, w1, h2)
        else:
            dw1, dc1 = ops.wgrad_bias(dpre, h2)
            dh2 = sgemm(dpre, w1, OP_N, OP_N)
        dx1, dg2, db2 = ops.layernorm_bwd(dh2, x1, m2, r2, g2, add=d2)        # + skip connection
        if pair & 2:
            da, dwo, dbo = ops.dgrad_wgrad(dx1, wo, a2)
        else:
            dwo, dbo = ops.wgrad_bias(dx1, a2)
            da = sgemm(dx1, wo, OP_N, OP_N)
        dqkv = torch.empty_like(qkv)
        q3, d3 = qkv.view(B, T, 3 * e), dqkv.view(B, T, 3 * e)
        ops.attention_bwd(q3[..., :e], q3[..., e:2 * e], q3[..., 2 * e:], None, heads, scale, a2.view(B, T, e), lse,
                          da.view(B, T, e), d3[..., :e], d3[..., e:2 * e], d3[..., 2 * e:])
        if pair & 1:
            dh1, dwqkv, dbqkv = ops.dgrad_wgrad(dqkv, wqkv, h1)
        else:
            dwqkv, dbqkv = ops.wgrad_bias(dqkv, h1)
            dh1 = sgemm(dqkv, wqkv, OP_N, OP_N)
        dx, dg1, db1 = ops.layernorm_bwd(dh1, x2, m1, r1, g1, add=dx1)          # + skip connection
        return (dx.view(B, T, e), None, None, dg1, db1, dwqkv, dbqkv, dwo, dbo, dg2, db2, dw1, dc1, dw2, dc2)


@_remember_precision
class _PlaneVitTrunk(torch.autograd.Function):
    """Blocks 0 .. n-1 of the build-defined pre-norm ViT as ONE node with every Linear product on the plane kernels
    (csrc/pgemm.hip: fp32-grade products on the bf16 matrix cores from operands RESIDENT as bf16 planes; ops.plane_count()
    = 3 planes / 6 products, or 2 / 3).  Same arithmetic plan as pre_norm_block -- residual adds, bias adds, GELU and GELU'
    in GEMM epilogues -- with each operand split into planes ONCE, by the kernel that produces it: LayerNorm forward writes
    planes only, the GELU epilogue writes the planes of its activation, LayerNorm backward and the GELU' epilogue write the
    planes of the gradients they produce (and the column sums that are the bias gradients); the attention backward writes
    the planes of dqkv (msn_attention_bwd_planes); the attention output is split by msn_plane_split; the weights and their transposes
    (for the input-gradient products) once per call.  The residual stream, LayerNorm statistics, attention and every
    parameter gradient stay fp32.

    Per-block parameter order: g1, b1, wqkv, bqkv, wo, bo, g2, b2, w1, c1, w2, c2."""

    NS = 10   # tensors saved per block

    @staticmethod
    def forward(ctx, x, heads, eps, n_blocks, need_grad, *P):
        # need_grad comes from the caller (plane_vit_trunk): ctx.needs_input_grad reflects requires_grad flags, not the grad mode, and
        # inside forward() the grad mode is always off -- under torch.no_grad() with trainable parameters it would stay True
        B, T, e = x.shape
        M = B * T
        NPL = ops.plane_count()
        scale = 1.0 / math.sqrt(e // heads)
        x2 = _c(x).view(M, e)
        saved, planes = [], []
        # the planes of every block's four weight matrices from one launch (44 launches of 5-6 us in the headline tower)
        wp = ops.plane_split_list([P[12 * i + k] for i in range(n_blocks) for k in (2, 4, 8, 10)], NPL)
        for i in range(n_blocks):
            g1, b1, wqkv, bqkv, wo, bo, g2, b2, w1, c1, w2, c2 = P[12 * i: 12 * i + 12]
            wqkvp, wop, w1p, w2p = wp[4 * i: 4 * i + 4]
            h1p, m1, r1 = ops.layernorm_fwd_planes(x2, g1, b1, eps, NPL)
            qkv = ops.pgemm_nt(h1p, wqkvp, bias=bqkv)
            q3 = qkv.view(B, T, 3 * e)
            if ops.attention_fwd_planes_supported(T, e // heads):     # the output projection's operand from the attention kernel
                a, lse, ap = ops.attention_fwd_planes(q3, heads, scale, NPL)
                a2 = a.view(M, e)
            else:
                a, lse = ops.attention_fwd(q3[..., :e], q3[..., e:2 * e], q3[..., 2 * e:], None, heads, scale)
                a2 = a.view(M, e)
                ap = ops.plane_split(a2, NPL)
            x1 = ops.pgemm_nt(ap, wop, bias=bo, epilogue=EPI_ADD, aux=x2)
            h2p, m2, r2 = ops.layernorm_fwd_planes(x1, g2, b2, eps, NPL)
            if need_grad:
                fp, dact = ops.pgemm_nt(h2p, w1p, bias=c1, epilogue=EPI_GELU, aux=True, out_planes=True)
            else:                                             # inference: no gelu' matrix, nothing kept
                fp = ops.pgemm_nt(h2p, w1p, bias=c1, epilogue=EPI_GELU, out_planes=True)
            out = ops.pgemm_nt(fp, w2p, bias=c2, epilogue=EPI_ADD, aux=x1)
            if need_grad:
                saved += [x2, m1, r1, qkv, a2, lse, x1, m2, r2, dact]
                planes.append((h1p, ap, h2p, fp))
            x2 = out
        ctx.dims = (B, T, e, heads, scale, n_blocks, NPL)
        if need_grad:
            ctx.planes = planes                               # plane matrices are not tensors: kept on the node
            ctx.save_for_backward(*saved, *P)
        return x2.view(B, T, e)

    @staticmethod
    def backward(ctx, dy):
        B, T, e, heads, scale, n_blocks, NPL = ctx.dims
        M = B * T
        t = ctx.saved_tensors
        NS = _PlaneVitTrunk.NS
        acts, P = t[:NS * n_blocks], t[NS * n_blocks:]
        grads = [None] * len(P)
        d2 = _c(dy).view(M, e)
        # the gradient entering the trunk is the one operand nobody produced in plane form; its column sums are the bias
        # gradient of the last block's second MLP layer (the other blocks get theirs from the LayerNorm backward above them)
        d2p, dc2 = ops.plane_split(d2, NPL, want_colsum=True)
        wt = ops.plane_split_list([P[12 * i + k] for i in range(n_blocks) for k in (2, 4, 8, 10)], NPL, transposed=True)
        for i in range(n_blocks - 1, -1, -1):
            wqkvt, wot, w1t, w2t = wt[4 * i: 4 * i + 4]
            x2, m1, r1, qkv, a2, lse, x1, m2, r2, dact = acts[NS * i: NS * i + NS]
            if ctx.planes[i] is None:
                raise RuntimeError("_PlaneVitTrunk: backward through this node a second time (retain_graph=True) is not supported: "
                                   "the bf16 plane operands of a block are released as soon as its gradients exist")
            h1p, ap, h2p, fp = ctx.planes[i]
            g1, b1, wqkv, bqkv, wo, bo, g2, b2, w1, c1, w2, c2 = P[12 * i: 12 * i + 12]
            # input gradients dX = dY . W are NT products against the planes of the transposed weight; every bias gradient (a
            # column sum of a gradient matrix) comes out of the kernel that writes that matrix's planes
            dw2 = ops.pgemm_tn(d2p, fp)
            dprep, dc1 = ops.pgemm_nt(d2p, w2t, epilogue=EPI_GELU_BWD, aux=dact,
                                      out_planes=True, want_colsum=True)
            dw1 = ops.pgemm_tn(dprep, h2p)
            dh2 = ops.pgemm_nt(dprep, w1t)
            dx1, dx1p, dg2, db2, dbo = ops.layernorm_bwd_planes(dh2, x1, m2, r2, g2, NPL, add=d2, want_colsum=True)   # + skip
            dwo = ops.pgemm_tn(dx1p, ap)
            da = ops.pgemm_nt(dx1p, wot)
            if ops.attention_bwd_planes_supported(T, e // heads):
                # one launch: Q, K, V, dO of a (sample, head) together in LDS, dqkv leaves as planes + column sums
                dqkvp, dbqkv = ops.attention_bwd_planes(qkv.view(B, T, 3 * e), heads, scale, a2.view(B, T, e), lse,
                                                        da.view(B, T, e), NPL)
            else:
                dqkv = torch.empty_like(qkv)
                q3, d3 = qkv.view(B, T, 3 * e), dqkv.view(B, T, 3 * e)
                ops.attention_bwd(q3[..., :e], q3[..., e:2 * e], q3[..., 2 * e:], None, heads, scale, a2.view(B, T, e), lse,
                                  da.view(B, T, e), d3[..., :e], d3[..., e:2 * e], d3[..., 2 * e:])
                dqkvp, dbqkv = ops.plane_split(dqkv, NPL, want_colsum=True)
            dwqkv = ops.pgemm_tn(dqkvp, h1p)
            dh1 = ops.pgemm_nt(dqkvp, wqkvt)
            grads[12 * i: 12 * i + 12] = [None, None, dwqkv, dbqkv, dwo, dbo, dg2, db2, dw1, dc1, dw2, dc2]
            d2, d2p, dg1, db1, dc2 = ops.layernorm_bwd_planes(dh1, x2, m1, r1, g1, NPL, add=dx1, want_colsum=True)    # + skip
            grads[12 * i], grads[12 * i + 1] = dg1, db1
            ctx.planes[i] = None                              # this block's planes are done with
        return (d2.view(B, T, e), None, None, None, None, *grads)


def plane_vit_trunk(x, heads, eps, block_params):
    """block_params: one 12-tuple (g1, b1, wqkv, bqkv, wo, bo, g2, b2, w1, c1, w2, c2) per block."""
    flat = [p for blk in block_params for p in blk]
    need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in flat))
    return _PlaneVitTrunk.apply(x, heads, eps, len(block_params), need_grad, *flat)


def plane_path_ok(x):
    """The plane kernels take the wide token matrices of the ViT towers (256-row tiles, 128-wide tiles)."""
    B, T, e = x.shape
    return bool(ops.plane_count()) and x.is_cuda and ops.pgemm_supported(B * T, e, e)


# Backward pairs (dX and dW of one Linear) as ONE work-list launch: bit 1 = the qkv and ff1 pairs (e-wide dX), bit 4 = the ff2
# pair (4e-wide dX), bit 2 = the e x e output projection.  None: by row count (_pair_backward); tests set the module attribute.
PAIR_BACKWARD = None


def _pair_backward(rows):
    """Which Linear backward pairs of a ViT block run as one work-list launch.  Measured on the headline step and
    tools/bench_gemm_list.py (profiles/r03_gemm_worklist_vs_flat.txt): with up to ~40 000 token rows (512 cutouts x 65 tokens)
    the launches of the qkv and ff1 pairs are under-filled and the pair wins 4-16 % over dgrad + wgrad + split-K sum; the
    ff2 pair (its dX has four times the tiles) wins up to ~12 000 rows (128 cutouts: 194 -> 182 us) and loses from 16 640 on
    (353 -> 372 us); at 66 560 rows the flat launches are full and every pair only ties; the e x e projection loses at every
    size (its two products are too short for the slabs of the cut tiles)."""
    if PAIR_BACKWARD is not None:
        return int(PAIR_BACKWARD)
    return (1 if rows <= 40000 else 0) | (4 if rows <= 12000 else 0)


def pre_norm_block(x, heads, p, eps=1e-6):
    if plane_path_ok(x):
        return plane_vit_trunk(x, heads, eps, [p])
    return _PreNormBlock.apply(x, heads, eps, *p)


@_remember_precision
class _PreNormLastBlock(torch.autograd.Function):
    """The LAST block of the build-defined ViT, evaluated for the class token only.  The head reads token 0 of the last
    block's output and nothing else, so of that block only the class row of x + Attn(LN1(x)) and of the MLP is ever used
    -- keys and values still come from every token.  Output (B, e) == pre_norm_block(x)[:, 0]; every gradient is identical
    (the rows left out have exactly zero gradient in the full evaluation).  Saves, per step, the query / output
    projections, the attention rows and the whole MLP of (T - 1) / T of the last block's tokens."""

    @staticmethod
    def forward(ctx, x, heads, eps, g1, b1, wqkv, bqkv, wo, bo, g2, b2, w1, c1, w2, c2):
        B, T, e = x.shape
        x3 = _c(x)
        x2 = x3.view(B * T, e)
        scale = 1.0 / math.sqrt(e // heads)
        h1, m1, r1 = ops.layernorm_fwd(x2, g1, b1, eps)
        kv = sgemm(h1, wqkv[e:], OP_N, OP_T, bias=bqkv[e:])                   # keys | values of every token
        h1c = h1.view(B, T, e)[:, 0, :]                                       # class rows (row stride T * e)
        q = sgemm(h1c, wqkv[:e], OP_N, OP_T, bias=bqkv[:e])                   # (B, e): the one query per sample
        cls_kernels = ops.cls_attention_supported(T, e // heads)                # one query per (sample, head): csrc/cls_attention.hip
        if cls_kernels:
            a2, lse = ops.cls_attention_fwd(q, kv, T, heads, scale)             # (lse: the probabilities here)
        else:
            kv3 = kv.view(B, T, 2 * e)
            a, lse = ops.attention_fwd(q.view(B, 1, e), kv3[..., :e], kv3[..., e:], None, heads, scale)
            a2 = a.view(B, e)
        x1 = sgemm(a2, wo, OP_N, OP_T, bias=bo, epilogue=EPI_ADD, aux=x3[:, 0, :])
        h2, m2, r2 = ops.layernorm_fwd(x1, g2, b2, eps)
        pre = torch.empty((B, w1.shape[0]), dtype=torch.float32, device=x.device)
        f = sgemm(h2, w1, OP_N, OP_T, bias=c1, epilogue=EPI_GELU, aux=pre)
        out = sgemm(f, w2, OP_N, OP_T, bias=c2, epilogue=EPI_ADD, aux=x1)
        ctx.dims = (B, T, e, heads, scale, cls_kernels)
        ctx.save_for_backward(x2, g1, wqkv, wo, g2, w1, w2, m1, r1, h1, kv, q, a2, lse, x1, m2, r2, h2, pre, f)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, T, e, heads, scale, cls_kernels = ctx.dims
        (x2, g1, wqkv, wo, g2, w1, w2, m1, r1, h1, kv, q, a2, lse, x1, m2, r2, h2, pre, f) = ctx.saved_tensors
        d = _c(dy)
        dw2, dc2 = ops.wgrad_bias(d, f)
        dpre = sgemm(d, w2, OP_N, OP_N, epilogue=EPI_GELU_BWD, aux=pre)
        dw1, dc1 = ops.wgrad_bias(dpre, h2)
        dh2 = sgemm(dpre, w1, OP_N, OP_N)
        dx1, dg2, db2 = ops.layernorm_bwd(dh2, x1, m2, r2, g2, add=d)         # + skip connection
        dwo, dbo = ops.wgrad_bias(dx1, a2)
        da = sgemm(dx1, wo, OP_N, OP_N)
        if cls_kernels:
            dq2, dkv = ops.cls_attention_bwd(q, kv, T, heads, scale, a2, lse, da)
        else:
            dq = torch.empty((B, 1, e), dtype=torch.float32, device=d.device)
            dkv = torch.empty_like(kv)
            kv3, d3 = kv.view(B, T, 2 * e), dkv.view(B, T, 2 * e)
            ops.attention_bwd(q.view(B, 1, e), kv3[..., :e], kv3[..., e:], None, heads, scale, a2.view(B, 1, e), lse,
                              da.view(B, 1, e), dq, d3[..., :e], d3[..., e:])
            dq2 = dq.view(B, e)
        h1c = h1.view(B, T, e)[:, 0, :]
        dwqkv = torch.empty_like(wqkv)
        dbqkv = torch.empty(3 * e, dtype=torch.float32, device=d.device)
        ops.wgrad_bias(dq2, h1c, out=(dwqkv[:e], dbqkv[:e]))
        ops.wgrad_bias(dkv, h1, out=(dwqkv[e:], dbqkv[e:]))
        dh1 = sgemm(dkv, wqkv[e:], OP_N, OP_N)
        dh1c = dh1.view(B, T, e)[:, 0, :]
        sgemm(dq2, wqkv[:e], OP_N, OP_N, epilogue=EPI_ADD, aux=dh1c, out=dh1c)   # the query branch reaches class rows only
        dx, dg1, db1 = ops.layernorm_bwd(dh1, x2, m1, r1, g1)
        ops.add_rows(dx.view(B, T, e)[:, 0, :], dx1)                          # skip connection of the class rows
        return (dx.view(B, T, e), None, None, dg1, db1, dwqkv, dbqkv, dwo, dbo, dg2, db2, dw1, dc1, dw2, dc2)


class _Bf16LastBlock(torch.autograd.Function):
    """_PreNormLastBlock for the bf16-RESIDENT tower (BASELINE cfg5; arithmetic of _Bf16VitTrunk below): the three products of
    the block that run over every token -- the key | value projection, its input gradient and its weight gradient -- take bf16
    operands from HBM on msn_bgemm_nt / msn_bgemm_tn (LayerNorm writes bf16, the projection writes bf16 keys | values, the
    class-token attention reads them and writes a bf16 gradient) instead of rounding fp32 operands inside the older 128 x 128
    kernel; the class-row products (B rows) stay where they were.  Same output and gradients as _PreNormLastBlock under the
    "bf16" GEMM precision up to the bf16 storage of keys | values and their gradient."""

    @staticmethod
    def forward(ctx, x, heads, eps, g1, b1, wqkv, bqkv, wo, bo, g2, b2, w1, c1, w2, c2):
        B, T, e = x.shape
        x3 = _c(x)
        x2 = x3.view(B * T, e)
        scale = 1.0 / math.sqrt(e // heads)
        h1, m1, r1 = ops.layernorm_fwd_bf16(x2, g1, b1, eps)
        kv = ops.bgemm_nt(h1, ops.cast_bf16(wqkv[e:]), bias=bqkv[e:], out_bf16=True)   # keys | values of every token, bf16
        h1c = h1.view(B, T, e)[:, 0, :].float()                               # class rows
        q = sgemm(h1c, wqkv[:e], OP_N, OP_T, bias=bqkv[:e])                   # (B, e): the one query per sample
        a2, probs = ops.cls_attention_fwd(q, kv, T, heads, scale)
        x1 = sgemm(a2, wo, OP_N, OP_T, bias=bo, epilogue=EPI_ADD, aux=x3[:, 0, :])
        h2, m2, r2 = ops.layernorm_fwd(x1, g2, b2, eps)
        pre = torch.empty((B, w1.shape[0]), dtype=torch.float32, device=x.device)
        f = sgemm(h2, w1, OP_N, OP_T, bias=c1, epilogue=EPI_GELU, aux=pre)
        out = sgemm(f, w2, OP_N, OP_T, bias=c2, epilogue=EPI_ADD, aux=x1)
        ctx.dims = (B, T, e, heads, scale)
        ctx.save_for_backward(x2, g1, wqkv, wo, g2, w1, w2, m1, r1, h1, kv, q, a2, probs, x1, m2, r2, h2, pre, f)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, T, e, heads, scale = ctx.dims
        (x2, g1, wqkv, wo, g2, w1, w2, m1, r1, h1, kv, q, a2, probs, x1, m2, r2, h2, pre, f) = ctx.saved_tensors
        d = _c(dy)
        dw2, dc2 = ops.wgrad_bias(d, f)
        dpre = sgemm(d, w2, OP_N, OP_N, epilogue=EPI_GELU_BWD, aux=pre)
        dw1, dc1 = ops.wgrad_bias(dpre, h2)
        dh2 = sgemm(dpre, w1, OP_N, OP_N)
        dx1, dg2, db2 = ops.layernorm_bwd(dh2, x1, m2, r2, g2, add=d)         # + skip connection
        dwo, dbo = ops.wgrad_bias(dx1, a2)
        da = sgemm(dx1, wo, OP_N, OP_N)
        dq2, dkv = ops.cls_attention_bwd(q, kv, T, heads, scale, a2, probs, da)   # dkv: bf16, the operand of the two products below
        h1c = h1.view(B, T, e)[:, 0, :].float()
        dwqkv = torch.empty_like(wqkv)
        dbqkv = torch.empty(3 * e, dtype=torch.float32, device=d.device)
        ops.wgrad_bias(dq2, h1c, out=(dwqkv[:e], dbqkv[:e]))
        dwqkv[e:].copy_(ops.bgemm_tn(dkv, h1))
        dbqkv[e:].copy_(ops.bcolsum(dkv))
        dh1 = ops.bgemm_nt(dkv, ops.cast_bf16_t(wqkv[e:]))                    # fp32: the class rows take the query branch below
        dh1c = dh1.view(B, T, e)[:, 0, :]
        sgemm(dq2, wqkv[:e], OP_N, OP_N, epilogue=EPI_ADD, aux=dh1c, out=dh1c)
        dx, dg1, db1 = ops.layernorm_bwd(dh1, x2, m1, r1, g1)
        ops.add_rows(dx.view(B, T, e)[:, 0, :], dx1)                          # skip connection of the class rows
        return (dx.view(B, T, e), None, None, dg1, db1, dwqkv, dbqkv, dwo, dbo, dg2, db2, dw1, dc1, dw2, dc2)


def pre_norm_last_block(x, heads, p, eps=1e-6, bf16_resident=False):
    """(B, T, e) -> (B, e): the class-token row of pre_norm_block(x, ...).  bf16_resident: the form for the bf16-resident tower."""
    B, T, e = x.shape
    if bf16_resident and e % 64 == 0 and ops.cls_attention_supported(T, e // heads):
        return _Bf16LastBlock.apply(x, heads, eps, *p)
    return _PreNormLastBlock.apply(x, heads, eps, *p)


BF16_DGRAD = True      # bf16-resident trunk (cfg5): input gradients that feed a LayerNorm backward are written as bf16


class _Bf16VitTrunk(torch.autograd.Function):
    """Blocks 0 .. n-1 of the build-defined pre-norm ViT as ONE node with bf16-RESIDENT GEMM operands (BASELINE cfg5:
    "ViT-B/16 bf16 on MFMA"): LayerNorm writes bf16, the GELU epilogue writes the bf16 activation and pre-activation, the
    weights are rounded once per call (plus a transposed copy for the input-gradient products), every product runs on
    msn_bgemm_nt / msn_bgemm_tn (256 x 256 tiles, LDS-DMA) with fp32 accumulation.  The residual stream, LayerNorm
    statistics, softmax and every parameter gradient stay fp32.  Arithmetic = the "bf16" GEMM precision of
    pre_norm_block (operands rounded to bf16), plus bf16 storage of the saved GELU pre-activation.

    Per-block parameter order: g1, b1, wqkv, bqkv, wo, bo, g2, b2, w1, c1, w2, c2."""

    NS = 14   # tensors saved per block

    @staticmethod
    def forward(ctx, x, heads, eps, n_blocks, *P):
        B, T, e = x.shape
        M = B * T
        scale = 1.0 / math.sqrt(e // heads)
        battn = ops.attention_bf16_supported(T, e // heads)     # 64-wide heads, T <= 256: attention on the bf16 matrix cores
        x2 = _c(x).view(M, e)
        saved = []
        empty = x2.new_empty(0)
        # every block's four weights rounded to bf16 by ONE launch (with the backward's transposed copies: 94 launches of 6 - 10 us per step)
        wb = ops.cast_bf16_list([_c(P[12 * i + j]) for i in range(n_blocks) for j in (2, 4, 8, 10)])
        for i in range(n_blocks):
            g1, b1, wqkv, bqkv, wo, bo, g2, b2, w1, c1, w2, c2 = P[12 * i: 12 * i + 12]
            wqkv_b, wo_b, w1_b, w2_b = wb[4 * i: 4 * i + 4]
            h1, m1, r1 = ops.layernorm_fwd_bf16(x2, g1, b1, eps)
            if battn:
                qkv = ops.bgemm_nt(h1, wqkv_b, bias=bqkv, out_bf16=True)
                ab, lse = ops.attention_bf16_fwd(qkv, B, T, heads, scale)
                a2 = empty
            else:
                qkv = ops.bgemm_nt(h1, wqkv_b, bias=bqkv)                             # fp32: the fp32 attention kernels' input
                q3 = qkv.view(B, T, 3 * e)
                a, lse = ops.attention_fwd(q3[..., :e], q3[..., e:2 * e], q3[..., 2 * e:], None, heads, scale)
                a2 = a.view(M, e)
                ab = ops.cast_bf16(a2)
            x1 = ops.bgemm_nt(ab, wo_b, bias=bo, epilogue=ops.BEPI_ADD, aux=x2)
            h2, m2, r2 = ops.layernorm_fwd_bf16(x1, g2, b2, eps)
            f, pre = ops.bgemm_nt(h2, w1_b, bias=c1, epilogue=ops.BEPI_GELU, out_bf16=True)
            out = ops.bgemm_nt(f, w2_b, bias=c2, epilogue=ops.BEPI_ADD, aux=x1)
            saved += [x2, m1, r1, h1, qkv, a2, ab, lse, x1, m2, r2, h2, pre, f]
            x2 = out
        ctx.dims = (B, T, e, heads, scale, n_blocks, battn)
        ctx.save_for_backward(*saved, *P)
        return x2.view(B, T, e)

    @staticmethod
    def backward(ctx, dy):
        B, T, e, heads, scale, n_blocks, battn = ctx.dims
        M = B * T
        t = ctx.saved_tensors
        NS = _Bf16VitTrunk.NS
        acts, P = t[:NS * n_blocks], t[NS * n_blocks:]
        grads = [None] * len(P)
        d2 = _c(dy).view(M, e)
        d2b = ops.cast_bf16(d2)
        dc2 = colsum(d2)                      # bias gradient of the last block's second MLP layer; the other blocks get
        # the transposed bf16 weight copies of every block from ONE launch
        wt = ops.cast_bf16_list([_c(P[12 * i + j]) for i in range(n_blocks) for j in (2, 4, 8, 10)], transposed=True)
        for i in range(n_blocks - 1, -1, -1):  # theirs from the LayerNorm backward that produced their d2
            x2, m1, r1, h1, qkv, a2, ab, lse, x1, m2, r2, h2, pre, f = acts[NS * i: NS * i + NS]
            g1, b1, wqkv, bqkv, wo, bo, g2, b2, w1, c1, w2, c2 = P[12 * i: 12 * i + 12]
            wqkv_t, wo_t, w1_t, w2_t = wt[4 * i: 4 * i + 4]
            # input gradients dX = dY . W are NT products against the transposed bf16 weight copy (K-contiguous operands);
            # every bias gradient (a column sum of a gradient matrix) comes out of the kernel that writes that matrix
            dw2 = ops.bgemm_tn(d2b, f)
            dpre, dc1 = ops.bgemm_nt(d2b, w2_t, epilogue=ops.BEPI_GELU_BWD, aux=pre, out_bf16=True,
                                     want_colsum=True)
            dw1 = ops.bgemm_tn(dpre, h2)
            # (the gradients entering the LayerNorm backwards leave their products as bf16 -- what autocast's dgrad gives: 155 MB less
            #  out of each epilogue and into each LayerNorm backward at 100 864 x 768; BF16_DGRAD = False: fp32)
            dh2 = ops.bgemm_nt(dpre, w1_t, out_bf16=BF16_DGRAD)
            dx1, dx1b, dg2, db2, dbo = ops.layernorm_bwd_bf16(dh2, x1, m2, r2, g2, add=d2, want_colsum=True)   # + skip
            dwo = ops.bgemm_tn(dx1b, ab)
            if battn:
                da = ops.bgemm_nt(dx1b, wo_t, out_bf16=True)
                dqkvb, dbqkv = ops.attention_bf16_bwd(qkv, ab, da, lse, B, T, heads, scale, want_colsum=True)
            else:
                da = ops.bgemm_nt(dx1b, wo_t)
                dqkv = torch.empty_like(qkv)
                q3, d3 = qkv.view(B, T, 3 * e), dqkv.view(B, T, 3 * e)
                ops.attention_bwd(q3[..., :e], q3[..., e:2 * e], q3[..., 2 * e:], None, heads, scale, a2.view(B, T, e), lse,
                                  da.view(B, T, e), d3[..., :e], d3[..., e:2 * e], d3[..., 2 * e:])
                dqkvb = ops.cast_bf16(dqkv)
                dbqkv = colsum(dqkv)
            dwqkv = ops.bgemm_tn(dqkvb, h1)
            dh1 = ops.bgemm_nt(dqkvb, wqkv_t, out_bf16=BF16_DGRAD)
            grads[12 * i: 12 * i + 12] = [None, None, dwqkv, dbqkv, dwo, dbo, dg2, db2, dw1, dc1, dw2, dc2]
            d2, d2b, dg1, db1, dc2 = ops.layernorm_bwd_bf16(dh1, x2, m1, r1, g1, add=dx1, want_colsum=True)   # + skip
            grads[12 * i], grads[12 * i + 1] = dg1, db1
        return (d2.view(B, T, e), None, None, None, *grads)


class _Bf16Linear(torch.autograd.Function):
    """y = x W^T + b with bf16-resident operands (msn_bgemm_nt / msn_bgemm_tn): the patch embedding of the bf16-resident tower --
    100 352 x 768 x 768 at cfg5, which the older 128 x 128 kernel ran from fp32 operands at a quarter of the rate."""

    @staticmethod
    def forward(ctx, x, w, b):
        shape = x.shape
        xb = ops.cast_bf16(_c(x).view(-1, shape[-1]))
        y = ops.bgemm_nt(xb, ops.cast_bf16(_c(w)), bias=b)
        ctx.save_for_backward(xb, w)
        ctx.shape, ctx.need_dx = shape, x.requires_grad
        return y.view(*shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xb, w = ctx.saved_tensors
        d = _c(dy).view(-1, dy.shape[-1])
        db16 = ops.cast_bf16(d)
        dw = ops.bgemm_tn(db16, xb)
        dx = ops.bgemm_nt(db16, ops.cast_bf16_t(_c(w))).view(ctx.shape) if ctx.need_dx else None
        return dx, dw, colsum(d)


def bf16_linear(x, weight, bias):
    """nn.Linear on the bf16-resident kernels (K % 64 == 0, N % 4 == 0, rows x K % 8 == 0); else `linear`."""
    K, N = x.shape[-1], weight.shape[0]
    if K % 64 == 0 and N % 4 == 0 and bias is not None and x.numel() % 8 == 0 and weight.numel() % 8 == 0:
        return _Bf16Linear.apply(x, weight, bias)
    return linear(x, weight, bias)


def bf16_vit_trunk(x, heads, eps, block_params):
    """block_params: one 12-tuple (g1, b1, wqkv, bqkv, wo, bo, g2, b2, w1, c1, w2, c2) per block."""
    flat = [p for blk in block_params for p in blk]
    return _Bf16VitTrunk.apply(x, heads, eps, len(block_params), *flat)


class _VitTokens(torch.autograd.Function):
    """[cls ; patch embeddings] + positional embedding."""

    @staticmethod
    def forward(ctx, patch_emb, cls, pos, B, T):
        return ops.vit_tokens_fwd(_c(patch_emb), _c(cls).view(-1), _c(pos).view(T, -1), B, T)

    @staticmethod
    def backward(ctx, dtok):
        dtok = _c(dtok)
        B, T, e = dtok.shape
        dpos = colsum(dtok.view(B, T * e)).view(1, T, e)
        dcls = dpos[:, :1, :].clone()
        return ops.vit_tokens_bwd(dtok), dcls, dpos, None, None


def vit_tokens(patch_emb, cls, pos, B, T):
    return _VitTokens.apply(patch_emb, cls, pos, B, T)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        y, m, r = ops.layernorm_fwd(_c(x), g, b, eps)
        ctx.save_for_backward(_c(x), m, r, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, m, r, g = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(_c(dy), x, m, r, g)
        return dx, dg, db, None


def layer_norm(x, g, b, eps=1e-5):
    return _LayerNorm.apply(x, g, b, eps)


class _Patchify(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, p):
        ctx.shape, ctx.p = tuple(img.shape), p
        return ops.patchify(_c(img.float()), p)

    @staticmethod
    def backward(ctx, dpatches):
        return ops.unpatchify(_c(dpatches), ctx.shape, ctx.p), None


def patchify(img, p):
    return _Patchify.apply(img, p)


class _TakeToken(torch.autograd.Function):
    """x[:, idx, :] of a (B, T, e) token tensor (the class token read-out) with a dense backward."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.shape, ctx.idx = tuple(x.shape), idx
        return x[:, idx, :].contiguous()

    @staticmethod
    def backward(ctx, dy):
        dx = torch.zeros(ctx.shape, dtype=dy.dtype, device=dy.device)
        dx[:, ctx.idx, :] = dy
        return dx, None


def take_token(x, idx):
    return _TakeToken.apply(x, idx)


# ------------------------------------------------------ channels-last convolution (build-defined encoders)
CONV_IMPLICIT = True   # False (tests, measurements): always im2col + GEMM
@_remember_precision
class _ConvCL(torch.autograd.Function):
    """y = conv(x) (+ bias) (+ ReLU) on channels-last tensors as an MFMA GEMM; weight keeps torch's (C_out, C_in, kh, kw)
    layout and is re-laid tap-major (C_out, kh, kw, C_in) per call, its gradient back.
    Channel counts that are multiples of 32 run as an IMPLICIT GEMM (msn_conv2d_fwd / _dgrad / _wgrad): the column matrix is
    never written, the GEMM's LDS-DMA lanes gather from the image and padding taps read a zero page; the backward
    keeps x instead of the 9x larger columns.  Other shapes (the 3-channel stem, 4-channel series features) and the
    dX of strided convolutions go through tap-major im2col / col2im (channels fastest: 16-byte channel groups); a
    channel count that is not a multiple of 4 is widened with zero channels first."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, relu):
        B, H, W, C = x.shape
        co, ci, kh, kw = weight.shape
        assert ci == C, f"conv expects {ci} input channels, got {C}"
        (sh, sw), (ph, pw) = stride, padding
        x = _c(x)
        plain = kh == 1 and kw == 1 and sh == 1 and sw == 1 and ph == 0 and pw == 0
        Cp = C if plain else (C + 3) // 4 * 4
        implicit = (CONV_IMPLICIT and (not plain) and ops._C_PRECISION[ops.GEMM_PRECISION] == ops.PREC_F32
                    and ops.conv2d_implicit_ok(B, H, W, C, co, kh, kw, sh, sw, ph, pw))
        oh, ow = ops.conv_out(H, kh, sh, ph), ops.conv_out(W, kw, sw, pw)
        if implicit:
            wmat = ops.conv_weight_relayout(_c(weight), co, ci, kh * kw, True)
            y = ops.conv2d_fwd(x, wmat, kh, kw, sh, sw, ph, pw, bias=bias, relu=relu)
            cols = x                                          # the backward gathers from the image again
        else:
            if plain:
                cols, wmat = x.view(B * H * W, C), weight.view(co, -1)
            else:
                xp = x if Cp == C else ops.pad_channels(x, Cp)
                cols = ops.im2col_tap(xp, kh, kw, sh, sw, ph, pw)
                wmat = ops.conv_weight_relayout(_c(weight), co, ci, kh * kw, True, ci_pad=Cp)
            y = sgemm(cols, wmat, OP_N, OP_T, bias=bias, epilogue=EPI_RELU if relu else EPI_NONE)
        ctx.geom = (B, H, W, C, Cp, kh, kw, sh, sw, ph, pw, plain, relu, bias is not None, tuple(weight.shape), implicit)
        ctx.save_for_backward(cols, wmat, y if relu else torch.empty(0), weight if implicit else torch.empty(0))
        return y.view(B, oh, ow, co)

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C, Cp, kh, kw, sh, sw, ph, pw, plain, relu, has_bias, wshape, implicit = ctx.geom
        cols, wmat, y, weight = ctx.saved_tensors
        co = wshape[0]
        d = _c(dy).view(-1, co)
        if relu:
            d = ops.relu_mask(d, y)
        if implicit:
            dw, db = ops.conv2d_wgrad(d, cols, kh, kw, sh, sw, ph, pw, want_bias=has_bias)
            dw = ops.conv_weight_relayout(dw, co, C, kh * kw, False).view(wshape)
            dx = None
            if ctx.needs_input_grad[0]:
                if sh == 1 and sw == 1:
                    dx = ops.conv2d_dgrad(d, ops.conv_weight_relayout(_c(weight), co, C, kh * kw, 2), (B, H, W, C), kh, kw, ph, pw)
                else:   # strided: 3/4 of the (input pixel, tap) pairs of a gather would be empty -- scatter through columns
                    dx = ops.col2im_tap(sgemm(d, wmat, OP_N, OP_N), (B, H, W, C), kh, kw, sh, sw, ph, pw)
            return dx, dw, db, None, None, None
        if has_bias:
            dw, db = ops.wgrad_bias(d, cols)
        else:
            dw, db = sgemm(d, cols, OP_T, OP_N), None
        if not plain:
            dw = ops.conv_weight_relayout(dw, co, C, kh * kw, False, ci_pad=Cp)
        dw = dw.view(wshape)
        dx = None
        if ctx.needs_input_grad[0]:
            dcols = sgemm(d, wmat, OP_N, OP_N)
            if plain:
                dx = dcols.view(B, H, W, C)
            else:
                dx = ops.col2im_tap(dcols, (B, H, W, Cp), kh, kw, sh, sw, ph, pw)
                if Cp != C:
                    dx = dx[..., :C].contiguous()    # gradient of the zero channels is dropped (data movement only)
        return dx, dw, db, None, None, None


def conv_cl(x, weight, bias=None, stride=(1, 1), padding=(0, 0), relu=False):
    return _ConvCL.apply(x, weight, bias, tuple(stride), tuple(padding), relu)


class _BatchNormAct(torch.autograd.Function):
    """y = [relu](BatchNorm(x) [+ residual]) over the rows of a (rows, C) token matrix (channels-last BN2d)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, residual, relu):
        shape = x.shape
        x2 = _c(x).view(-1, shape[-1])
        r2 = _c(residual).view(-1, shape[-1]) if residual is not None else None
        y, mean, rstd = ops.batchnorm_fwd(x2, gamma, beta, running_mean, running_var, training, residual=r2, relu=relu)
        ctx.cfg = (bool(training), bool(relu), residual is not None, shape)
        ctx.save_for_backward(x2, gamma, mean, rstd, y if relu else torch.empty(0))
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        training, relu, has_res, shape = ctx.cfg
        x2, gamma, mean, rstd, y = ctx.saved_tensors
        d = _c(dy).view(-1, shape[-1])
        fused_mask = relu and not has_res and not (training and ops._bn_sync_world() > 1)
        if relu and not fused_mask:       # the masked gradient is also the residual branch's gradient / the all-reduced sums' input
            d = ops.relu_mask(d, y)
        dx, dg, db = ops.batchnorm_bwd(d, x2, None, mean, rstd, gamma, training, relu_out=y.view(-1, shape[-1]) if fused_mask else None)
        return dx.view(shape), dg, db, None, None, None, (d.view(shape) if has_res else None), None


BN_COUNTERS = None   # a list while an encoder collects the step counters of its BatchNorms (bump_batchnorm_counters)


def batchnorm_act(x, bn, training, residual=None, relu=False):
    """`bn` is an nn.BatchNorm2d parameter holder."""
    out = _BatchNormAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, residual, relu)
    if training:
        if BN_COUNTERS is not None:
            BN_COUNTERS.append(bn.num_batches_tracked)
        else:
            with torch.no_grad():
                bn.num_batches_tracked += 1
    return out


class collect_batchnorm_counters:
    """`with collect_batchnorm_counters():` around a forward pass: the `num_batches_tracked += 1` of every BatchNorm inside becomes ONE
    fused add at the end (ResNet-18: 20 one-element launches per step otherwise)."""

    def __enter__(self):
        global BN_COUNTERS
        self.prev, BN_COUNTERS = BN_COUNTERS, []
        return self

    def __exit__(self, *exc):
        global BN_COUNTERS
        counters, BN_COUNTERS = BN_COUNTERS, self.prev
        if counters and exc[0] is None:
            with torch.no_grad():
                torch._foreach_add_(counters, 1)
        return False


class _MaxPoolCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        y, arg = ops.maxpool2d_fwd(_c(x), k, s, p)
        ctx.cfg = (tuple(x.shape), k, s, p)
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        shape, k, s, p = ctx.cfg
        (arg,) = ctx.saved_tensors
        return ops.maxpool2d_bwd(_c(dy), arg, shape, k, s, p), None, None, None


def maxpool_cl(x, k, s, p):
    return _MaxPoolCL.apply(x, k, s, p)


def to_channels_last(img):
    """(B, C, H, W) -> (B, H, W, C) through the 1x1 patch gather (and its adjoint in backward)."""
    B, C, H, W = img.shape
    return patchify(img, 1).view(B, H, W, C)

"""GPU parity of the row-wise kernels and the fused attention (raw C-ABI level) against plain
torch fp64 on the same inputs.  Tolerance 1e-4 relative unless noted (north star: 1e-3)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-5)


def _g(seed):
    return torch.Generator().manual_seed(seed)


# the last three are token-matrix sizes of the towers: the backward's block partials fill all 512 workgroups there
@pytest.mark.parametrize("rows,cols", [(1, 8), (37, 16), (1000, 32), (513, 64), (300, 384), (65, 768), (9, 1024),
                                       (66560, 384), (204800, 64), (50001, 12)])
def test_layernorm_fwd_bwd(rows, cols):
    from multimodal_supernovae_amd import ops
    g = _g(rows + cols)
    x = torch.randn(rows, cols, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(cols, generator=g) + 1, torch.randn(cols, generator=g)
    dy = torch.randn(rows, cols, generator=g)
    xr = x.double().requires_grad_()
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    yr = F.layer_norm(xr, (cols,), gr, br, 1e-5)
    yr.backward(dy.double())
    y, mean, rstd = ops.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda())
    torch.testing.assert_close(y.cpu().double(), yr.detach(), **TOL)
    dx, dg, db = ops.layernorm_bwd(dy.cuda(), x.cuda(), mean, rstd, gamma.cuda())
    torch.testing.assert_close(dx.cpu().double(), xr.grad, **TOL)
    col_tol = dict(rtol=1e-4, atol=1e-4 * max(1.0, rows ** 0.5 / 30))      # sums over `rows` terms
    torch.testing.assert_close(dg.cpu().double(), gr.grad, **col_tol)
    torch.testing.assert_close(db.cpu().double(), br.grad, **col_tol)
    add = torch.randn(rows, cols, generator=g)                              # residual-branch gradient fused in
    dx2, _, _ = ops.layernorm_bwd(dy.cuda(), x.cuda(), mean, rstd, gamma.cuda(), add=add.cuda())
    torch.testing.assert_close(dx2.cpu().double(), xr.grad + add.double(), **TOL)


@pytest.mark.parametrize("rows,cols", [(1, 8), (6, 16), (256, 128), (1000, 128), (77, 64)])
def test_l2norm_fwd_bwd(rows, cols):
    from multimodal_supernovae_amd import ops
    g = _g(rows * 3 + cols)
    x, dy = torch.randn(rows, cols, generator=g), torch.randn(rows, cols, generator=g)
    xr = x.double().requires_grad_()
    yr = xr / xr.norm(dim=-1, keepdim=True)
    yr.backward(dy.double())
    y, inv = ops.l2norm_fwd(x.cuda())
    torch.testing.assert_close(y.cpu().double(), yr.detach(), **TOL)
    dx = ops.l2norm_bwd(dy.cuda(), y, inv)
    torch.testing.assert_close(dx.cpu().double(), xr.grad, **TOL)


@pytest.mark.parametrize("B,T,e,nband", [(3, 12, 16, 1), (3, 12, 16, 2), (5, 200, 64, 2), (2, 1024, 32, 1),
                                        (4, 30, 384, 3), (2, 10, 6, 1)])
def test_time_embed_fwd_bwd(B, T, e, nband):
    from multimodal_supernovae_amd import ops
    from oracle.encoders import time_positional_encoding
    g = _g(B + T + e)
    x = torch.randn(B, T, generator=g)
    t = torch.rand(B, T, generator=g) * (9000.0 if nband == 1 else 100.0)
    w, bw = torch.randn(e, generator=g), torch.randn(e, generator=g)
    band = torch.randn(nband, e, generator=g) if nband > 1 else None
    norm = 17945.14
    omega = torch.exp(torch.arange(0, e, 2).float() * (-math.log(norm) / e))
    wr, bwr = w.clone().requires_grad_(), bw.clone().requires_grad_()
    ref = x[:, :, None] * wr + bwr + time_positional_encoding(t, e, norm)
    if nband > 1:
        bandr = band.clone().requires_grad_()
        ref = ref + bandr[torch.arange(nband).repeat_interleave(T // nband)][None]
    out = ops.time_embed_fwd(x.cuda(), t.cuda(), w.cuda(), bw.cuda(), omega.cuda(),
                             band.cuda() if band is not None else None)
    # sin/cos of arguments up to 9000 rad: fp32 argument rounding (1 ulp of t*omega ~ 5e-4) bounds parity
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-4, atol=2e-3 if nband == 1 else 5e-5)
    dy = torch.randn(B, T, e, generator=g)
    ref.backward(dy)
    dw, dbw, dband = ops.time_embed_bwd(dy.cuda(), x.cuda(), nband)
    torch.testing.assert_close(dw.cpu(), wr.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dbw.cpu(), bwr.grad, rtol=1e-4, atol=1e-3)
    if nband > 1:
        torch.testing.assert_close(dband.cpu(), bandr.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("mode", ["mean", "max"])
@pytest.mark.parametrize("B,T,e", [(3, 12, 16), (8, 200, 64), (2, 1024, 32), (4, 65, 384)])
def test_masked_pool(mode, B, T, e):
    from multimodal_supernovae_amd import ops
    g = _g(B * T + e)
    x = torch.randn(B, T, e, generator=g)
    mask = torch.rand(B, T, generator=g) > 0.4
    mask[:, 0] = True
    xr = x.clone().requires_grad_()
    z = xr * mask[:, :, None]
    ref = z.sum(1) / mask.sum(1)[:, None] if mode == "mean" else z.max(dim=1)[0]
    m = ops.POOL_MEAN if mode == "mean" else ops.POOL_MAX
    mu8 = ops._mask_u8(mask.cuda())
    out, arg, cnt = ops.masked_pool_fwd(x.cuda(), mu8, m)
    torch.testing.assert_close(out.cpu(), ref.detach(), **TOL)
    dout = torch.randn(B, e, generator=g)
    ref.backward(dout)
    dx = ops.masked_pool_bwd(dout.cuda(), mu8, T, m, arg, cnt)
    torch.testing.assert_close(dx.cpu(), xr.grad, **TOL)


def test_mean_pool_of_fully_padded_row_is_nan_like_the_reference():
    from multimodal_supernovae_amd import ops
    x = torch.randn(2, 5, 8)
    mask = torch.tensor([[True] * 5, [False] * 5])
    out, _, _ = ops.masked_pool_fwd(x.cuda(), ops._mask_u8(mask.cuda()), ops.POOL_MEAN)
    assert torch.isnan(out[1]).all() and not torch.isnan(out[0]).any()


def _attn_ref(q, k, v, mask, heads, scale):
    B, Tq, E = q.shape
    Tk = k.shape[1]
    s = E // heads
    qh, kh, vh = (t.view(B, -1, heads, s) for t in (q, k, v))
    dot = torch.einsum("bihs,bjhs->bhij", qh, kh) * scale
    if mask is not None:
        dot = torch.where(mask[:, None, None, :], dot, torch.full_like(dot, -1e7))
    p = torch.softmax(dot, dim=-1)
    return torch.einsum("bhij,bjhs->bihs", p, vh).reshape(B, Tq, E)


@pytest.fixture
def attention_path():
    """Select the vector-ALU (1) or matrix-core (2) attention kernels for one test, then restore auto."""
    from multimodal_supernovae_amd import _lib

    def choose(mode):
        _lib.check(_lib.lib().msn_set_attention_path(mode))
    yield choose
    _lib.lib().msn_set_attention_path(0)


@pytest.mark.parametrize("B,T,E,heads", [(3, 12, 16, 4), (2, 200, 64, 8), (2, 220, 32, 2), (1, 1024, 32, 2),
                                        (2, 65, 384, 6), (2, 33, 24, 2), (1, 300, 96, 3), (1, 197, 768, 12),
                                        (3, 80, 96, 2), (2, 256, 128, 4), (2, 16, 32, 2)])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("path", [1, 2])
def test_attention_fwd_bwd_packed_qkv(B, T, E, heads, masked, path, attention_path):
    """q|k|v live in one (B, T, 3E) buffer (the layout the transformer block uses).  path 1 = vector-ALU
    kernels, path 2 = matrix-core kernels wherever they apply (head width % 16 == 0, T <= 256)."""
    from multimodal_supernovae_amd import ops
    attention_path(path)
    g = _g(B * T + E + heads)
    qkv = torch.randn(B, T, 3 * E, generator=g)
    dout = torch.randn(B, T, E, generator=g)
    mask = None
    if masked:
        mask = torch.rand(B, T, generator=g) > 0.3
        mask[:, 0] = True
        mask[-1] = False           # a fully padded sample: uniform attention over the -1e7 fills
    scale = 1.0 / math.sqrt(E)
    r = qkv.double().requires_grad_()
    ref = _attn_ref(r[..., :E], r[..., E:2 * E], r[..., 2 * E:], mask, heads, scale)
    ref.backward(dout.double())
    dev = qkv.cuda()
    q, k, v = dev[..., :E], dev[..., E:2 * E], dev[..., 2 * E:]
    mu8 = ops._mask_u8(mask.cuda()) if masked else None
    out, lse = ops.attention_fwd(q, k, v, mu8, heads, scale)
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=1e-4, atol=2e-5)
    dqkv = torch.empty_like(dev)
    ops.attention_bwd(q, k, v, mu8, heads, scale, out, lse, dout.cuda(), dqkv[..., :E], dqkv[..., E:2 * E],
                      dqkv[..., 2 * E:])
    torch.testing.assert_close(dqkv.cpu().double(), r.grad, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("T,E", [(50, 16), (50, 64), (50, 128), (50, 256), (300, 128), (300, 256), (50, 140), (50, 36)])
def test_attention_single_shared_query(T, E):
    """The attn-pooling shape: one learnable query shared by the batch (q_bstride = 0), no mask, scale 1/sqrt(head_dim).
    Heads up to 32 wide run on the vector-ALU kernels, wider ones (emb 128 / 256 of ref transformer_utils.py:197-204 with its 2
    pooling heads = 64- / 128-wide heads) on the matrix cores, below and above their 128-token limit; 70- and 18-wide heads exercise
    the binding's zero-column padding with a shared query."""
    from multimodal_supernovae_amd import ops
    g = _g(77 + T + E)
    B, heads = 4, 2
    q = torch.randn(1, 1, E, generator=g)
    k, v = torch.randn(B, T, E, generator=g), torch.randn(B, T, E, generator=g)
    dout = torch.randn(B, 1, E, generator=g)
    scale = 1.0 / math.sqrt(E // heads)
    qr, kr, vr = q.double().requires_grad_(), k.double().requires_grad_(), v.double().requires_grad_()
    ref = _attn_ref(qr.expand(B, 1, E), kr, vr, None, heads, scale)
    ref.backward(dout.double())
    out, lse = ops.attention_fwd(q.cuda(), k.cuda(), v.cuda(), None, heads, scale, q_shared=True)
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=1e-4, atol=2e-5)
    dq = torch.empty(B, 1, E).cuda()
    dk, dv = torch.empty(B, T, E).cuda(), torch.empty(B, T, E).cuda()
    ops.attention_bwd(q.cuda(), k.cuda(), v.cuda(), None, heads, scale, out, lse, dout.cuda(), dq, dk, dv,
                      q_shared=True)
    torch.testing.assert_close(dq.sum(0, keepdim=True).cpu().double(), qr.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(dk.cpu().double(), kr.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(dv.cpu().double(), vr.grad, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("rows,C", [(7, 8), (1000, 32), (65536, 64), (200001, 130), (300000, 4)])
@pytest.mark.parametrize("training", [True, False])
def test_batchnorm_kernels_direct(rows, C, training):
    """msn_batchnorm_fwd / _bwd on channels-last (rows, C) matrices against torch's batch_norm in fp64, from a handful
    of rows to the many-block regime where the per-channel partials are summed by 16 thread groups; running statistics
    (momentum 0.1, unbiased variance) included."""
    from multimodal_supernovae_amd import ops
    g = _g(rows + C)
    x = torch.randn(rows, C, generator=g) * 1.5 + 0.3
    gamma, beta = torch.randn(C, generator=g) + 1, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    dy = torch.randn(rows, C, generator=g)
    xr, gr, br = x.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    yr = F.batch_norm(xr, rm_ref, rv_ref, gr, br, training=training, momentum=0.1, eps=1e-5)
    yr.backward(dy.double())
    rm_d, rv_d = rm.cuda(), rv.cuda()
    y, mean, rstd = ops.batchnorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), rm_d, rv_d, training)
    torch.testing.assert_close(y.cpu().double(), yr.detach(), rtol=1e-4, atol=2e-5)
    if training and rows > 1:
        torch.testing.assert_close(rm_d.cpu().double(), rm_ref, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rv_d.cpu().double(), rv_ref, rtol=1e-4, atol=1e-6)
    dx, dg, db = ops.batchnorm_bwd(dy.cuda(), x.cuda(), None, mean, rstd, gamma.cuda(), training)
    col = dict(rtol=1e-4, atol=1e-4 * max(1.0, rows ** 0.5 / 30))
    torch.testing.assert_close(dx.cpu().double(), xr.grad, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(dg.cpu().double(), gr.grad, **col)
    torch.testing.assert_close(db.cpu().double(), br.grad, **col)

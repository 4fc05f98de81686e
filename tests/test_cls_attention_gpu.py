"""msn_cls_attention_fwd / _bwd (csrc/cls_attention.hip): one query per (sample, head) over T keys -- the class-token row of the
build-defined ViT's last block.  Reference: the reference's attention formula (src/transformer_utils.py:36-89 for a single
query row) in fp64; fp32 keys | values to fp32 rounding, bf16 keys | values against the same formula on the bf16 values."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, kv, B, T, heads, scale):
    e = q.shape[1]
    hd = e // heads
    k, v = kv.view(B, T, 2 * e)[..., :e], kv.view(B, T, 2 * e)[..., e:]
    qh, kh, vh = q.view(B, heads, hd), k.reshape(B, T, heads, hd), v.reshape(B, T, heads, hd)
    s = torch.einsum("bhd,bthd->bht", qh, kh) * scale
    p = torch.softmax(s, dim=-1)
    return torch.einsum("bht,bthd->bhd", p, vh).reshape(B, e), p


@pytest.mark.parametrize("B,T,heads", [(3, 65, 6), (2, 197, 12), (1, 1, 2), (5, 256, 1), (4, 9, 3)])
@pytest.mark.parametrize("bf16", [False, True])
def test_cls_attention_matches_fp64(B, T, heads, bf16):
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(17 * T + heads)
    e = heads * 64
    q = torch.randn(B, e, generator=g)
    kv = torch.randn(B * T, 2 * e, generator=g)
    dout = torch.randn(B, e, generator=g)
    if bf16:
        kv = kv.to(torch.bfloat16)
    scale = 1.0 / math.sqrt(64)
    qr = q.double().requires_grad_()
    kvr = kv.double().requires_grad_()
    out_ref, p_ref = _ref(qr, kvr, B, T, heads, scale)
    out_ref.backward(dout.double())
    qc, kvc, dc = q.cuda(), kv.cuda(), dout.cuda()
    assert ops.cls_attention_supported(T, 64)
    out, probs = ops.cls_attention_fwd(qc, kvc, T, heads, scale)
    torch.testing.assert_close(out.cpu().double(), out_ref.detach(), rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(probs.cpu().double(), p_ref.detach(), rtol=2e-5, atol=1e-7)
    dq, dkv = ops.cls_attention_bwd(qc, kvc, T, heads, scale, out, probs, dc)
    torch.testing.assert_close(dq.cpu().double(), qr.grad, rtol=3e-5, atol=3e-6)
    assert dkv.dtype == kv.dtype
    if bf16:       # the gradient is stored as bf16: half an ulp of 8 significant bits
        torch.testing.assert_close(dkv.cpu().double(), kvr.grad, rtol=2.0 ** -8, atol=1e-6)
    else:
        torch.testing.assert_close(dkv.cpu().double(), kvr.grad, rtol=3e-5, atol=3e-6)


def test_cls_attention_strided_query_rows_and_determinism():
    """The query rows are the class rows of a (B, T, e) matrix (row stride T e); two launches give the same bits."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, heads = 4, 65, 6
    e = heads * 64
    h = torch.randn(B, T, e, generator=g).cuda()
    kv = torch.randn(B * T, 2 * e, generator=g).cuda()
    q = h[:, 0, :]
    out1, p1 = ops.cls_attention_fwd(q, kv, T, heads, 0.125)
    out2, p2 = ops.cls_attention_fwd(q.contiguous(), kv, T, heads, 0.125)
    assert torch.equal(out1, out2) and torch.equal(p1, p2)
    dout = torch.randn(B, e, generator=g).cuda()
    a = ops.cls_attention_bwd(q, kv, T, heads, 0.125, out1, p1, dout)
    b = ops.cls_attention_bwd(q.contiguous(), kv, T, heads, 0.125, out1, p1, dout)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_cls_attention_rejects_other_shapes():
    from multimodal_supernovae_amd import _lib, ops
    assert not ops.cls_attention_supported(300, 64) and not ops.cls_attention_supported(65, 32)
    q = torch.zeros(2, 64, device="cuda")
    kv = torch.zeros(2 * 300, 128, device="cuda")
    with pytest.raises(_lib.MsnHipError):
        ops.cls_attention_fwd(q, kv, 300, 1, 1.0)

"""One-pass self-attention backward (mattn_bwd_fused_kernel): against the dQ + dK,dV kernel pair it replaces (same products in
the same order; the two compilations contract multiply-adds differently, so equal to a few ulp, not bit for bit),
and its plane-output form (msn_attention_bwd_planes) against msn_attention_bwd + msn_plane_split on the same inputs.
The fp64 reference of the arithmetic itself is tests/test_attention_fuzz_gpu.py, whose self-attention cases up to 128
tokens run on this kernel by default."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(B, T, heads, hd, seed, masked):
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(seed)
    E = heads * hd
    qkv = torch.randn(B, T, 3 * E, generator=g).cuda()
    dout = torch.randn(B, T, E, generator=g).cuda()
    mu8 = None
    if masked:
        mask = torch.rand(B, T, generator=g) > 0.3
        mask[:, 0] = True
        mask[0, -1] = False
        if B > 1:
            mask[-1] = False                       # a fully padded sample
        mu8 = ops._mask_u8(mask.cuda())
    return qkv, dout, mu8


def _close(one, two):
    torch.testing.assert_close(one, two, rtol=2e-5, atol=2e-6 * two.abs().max().item())


def _bwd(qkv, dout, mu8, heads, scale, fused):
    from multimodal_supernovae_amd import ops
    E = dout.shape[-1]
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    out, lse = ops.attention_fwd(q, k, v, mu8, heads, scale)
    dqkv = torch.full_like(qkv, float("nan"))
    ops.set_attention_fused(fused)
    try:
        ops.attention_bwd(q, k, v, mu8, heads, scale, out, lse, dout, dqkv[..., :E], dqkv[..., E:2 * E], dqkv[..., 2 * E:])
    finally:
        ops.set_attention_fused(True)
    return out, lse, dqkv


@pytest.mark.parametrize("T", [5, 16, 17, 64, 65, 100, 113, 128])
@pytest.mark.parametrize("hd,heads", [(64, 6), (32, 2), (16, 3), (48, 1), (24, 2)])
@pytest.mark.parametrize("masked", [False, True])
def test_one_pass_equals_two_kernels_self_comparison(T, hd, heads, masked):
    qkv, dout, mu8 = _inputs(3, T, heads, hd, 77 * T + hd, masked)
    scale = 1.0 / math.sqrt(hd)
    _, _, two = _bwd(qkv, dout, mu8, heads, scale, fused=False)
    _, _, one = _bwd(qkv, dout, mu8, heads, scale, fused=True)
    assert not torch.isnan(one).any()
    _close(one, two)
    if T <= 65:       # up to four full tiles the default shares one recomputation between dQ and dK,dV: the other form too
        _, _, seven = _bwd(qkv, dout, mu8, heads, scale, fused=3)
        _close(seven, two)


def test_narrow_heads_on_the_matrix_cores():
    """8-wide heads run as 16-wide ones when the matrix-core path is forced (the light-curve transformer's shape, cut to
    the one-pass kernel's 128 tokens)."""
    from multimodal_supernovae_amd import _lib
    qkv, dout, mu8 = _inputs(4, 100, 8, 8, 5, True)
    _lib.check(_lib.lib().msn_set_attention_path(2))
    try:
        _, _, two = _bwd(qkv, dout, mu8, 8, 0.125, fused=False)
        _, _, one = _bwd(qkv, dout, mu8, 8, 0.125, fused=True)
    finally:
        _lib.lib().msn_set_attention_path(0)
    _close(one, two)


@pytest.mark.parametrize("planes", [3, 2])
@pytest.mark.parametrize("B,T,hd,heads", [(8, 65, 64, 6), (3, 65, 64, 6), (5, 17, 16, 2), (2, 128, 32, 3), (16, 64, 48, 2),
                                          (3, 33, 16, 3), (4, 65, 48, 1)])   # the last two: 3 E % 32 == 16 (a padding column block)
@pytest.mark.parametrize("masked", [False, True])
def test_plane_output_equals_split_of_fp32_gradient(planes, B, T, hd, heads, masked):
    """dqkv as planes == msn_plane_split of the fp32 dqkv, byte for byte (incl. the zero rows behind a matrix whose row
    count is not a multiple of 32); its column sums == those of the split pass up to the order of the additions."""
    from multimodal_supernovae_amd import ops
    qkv, dout, mu8 = _inputs(B, T, heads, hd, 31 * B + T, masked)
    E = heads * hd
    scale = 1.0 / math.sqrt(hd)
    out, lse, dqkv = _bwd(qkv, dout, mu8, heads, scale, fused=True)
    want, want_cs = ops.plane_split(dqkv.view(B * T, 3 * E), planes, want_colsum=True)
    # poison the destination's memory first: the kernel must write every byte it owns
    junk = ops.Planes.empty(B * T, 3 * E, planes, qkv.device)
    junk.buf.fill_(0x7F)
    del junk
    got, cs = ops.attention_bwd_planes(qkv, heads, scale, out, lse, dout, planes, want_colsum=True, mask_u8=mu8)
    assert torch.equal(got.buf, want.buf)
    ref_cs = dqkv.view(B * T, 3 * E).double().sum(0)
    tol = 1e-5 * dqkv.view(B * T, 3 * E).abs().double().sum(0).max().item()
    assert (cs.double() - ref_cs).abs().max().item() <= tol
    assert (want_cs.double() - ref_cs).abs().max().item() <= tol
    again = ops.attention_bwd_planes(qkv, heads, scale, out, lse, dout, planes, want_colsum=False, mask_u8=mu8)
    assert torch.equal(again.buf, want.buf)


def test_headline_shape_and_determinism():
    from multimodal_supernovae_amd import ops
    B, T, heads, hd = 64, 65, 6, 64
    qkv, dout, _ = _inputs(B, T, heads, hd, 9, False)
    scale = 1.0 / math.sqrt(hd)
    out, lse, dqkv = _bwd(qkv, dout, None, heads, scale, fused=True)
    a, ca = ops.attention_bwd_planes(qkv, heads, scale, out, lse, dout, 3)
    b, cb = ops.attention_bwd_planes(qkv, heads, scale, out, lse, dout, 3)
    assert torch.equal(a.buf, b.buf) and torch.equal(ca, cb)
    torch.testing.assert_close(a.to_float(), dqkv.view(B * T, -1), rtol=0, atol=1e-6 * dqkv.abs().max().item())


def test_rejects_what_it_cannot_do():
    from multimodal_supernovae_amd import ops
    from multimodal_supernovae_amd._lib import MsnHipError as MsnError
    qkv, dout, _ = _inputs(2, 200, 2, 64, 1, False)       # too long for the four images in LDS
    out, lse = torch.empty(2, 200, 128, device="cuda"), torch.empty(2, 2, 200, 2, device="cuda")
    with pytest.raises(MsnError):
        ops.attention_bwd_planes(qkv, 2, 0.125, out, lse, dout, 3)
    qkv, dout, _ = _inputs(2, 65, 2, 8, 1, False)         # heads narrower than a plane block
    out, lse = torch.empty(2, 65, 16, device="cuda"), torch.empty(2, 2, 65, 2, device="cuda")
    with pytest.raises(MsnError):
        ops.attention_bwd_planes(qkv, 2, 0.125, out, lse, dout, 3)


@pytest.mark.parametrize("B,T,H,hd", [(8, 65, 6, 64), (3, 33, 2, 32), (5, 64, 4, 16), (2, 128, 3, 64), (7, 17, 2, 64), (3, 65, 1, 16),
                                      (2, 40, 3, 16), (4, 1, 2, 64), (1024, 65, 6, 64)])
@pytest.mark.parametrize("planes", [3, 2])
def test_forward_writing_planes(B, T, H, hd, planes):
    """msn_attention_fwd_planes: out and lse are msn_attention_fwd's, bit for bit, and the plane matrix is msn_plane_split(out)'s bytes
    (rows of a sample start anywhere in a 32-row block; (B T) % 32 != 0 and E % 32 == 16 leave padding the call must zero)."""
    from multimodal_supernovae_amd import ops
    e = H * hd
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = torch.randn(B, T, 3 * e, generator=g).cuda()
    mask = (torch.rand(B, T, generator=g) > 0.2).to(torch.uint8).cuda() if T > 4 and B < 100 else None
    scale = 1.0 / math.sqrt(e)
    ref, lse_ref = ops.attention_fwd(qkv[..., :e], qkv[..., e:2 * e], qkv[..., 2 * e:], mask, H, scale)
    out, lse, op = ops.attention_fwd_planes(qkv, H, scale, planes, mask_u8=mask)
    assert torch.equal(out, ref) and torch.equal(lse, lse_ref)
    want = ops.plane_split(ref.view(B * T, e), planes)
    assert torch.equal(op.buf, want.buf)
    again = ops.attention_fwd_planes(qkv, H, scale, planes, mask_u8=mask)[2]      # into a fresh (uninitialised) buffer: padding written
    assert torch.equal(again.buf, want.buf)

"""Seeded fuzz of msn_attention_fwd / _bwd over head widths, sequence lengths (also Tq != Tk), masks and both kernel
families.  Reference: the reference's formula in fp64 (scores * scale, -1e7 key fill, softmax, @ v)."""
import math
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, mask, heads, scale):
    B, Tq, E = q.shape
    s = E // heads
    qh, kh, vh = (t.view(B, -1, heads, s) for t in (q, k, v))
    dot = torch.einsum("bihs,bjhs->bhij", qh, kh) * scale
    if mask is not None:
        dot = torch.where(mask[:, None, None, :], dot, torch.full_like(dot, -1e7))
    return torch.einsum("bhij,bjhs->bihs", torch.softmax(dot, dim=-1), vh).reshape(B, Tq, E)


@pytest.mark.parametrize("seed", range(5))
@pytest.mark.parametrize("path", [1, 2])
def test_attention_fuzz(seed, path):
    from multimodal_supernovae_amd import _lib, ops
    rng = random.Random(300 + seed)
    g = torch.Generator().manual_seed(seed)
    _lib.check(_lib.lib().msn_set_attention_path(path))
    try:
        for _ in range(14):
            heads = rng.choice([1, 2, 3, 4, 6, 8])
            hd = rng.choice([4, 8, 8, 12, 16, 16, 24, 32, 64])
            E = heads * hd
            B = rng.randint(1, 3)
            lengths = [1, 2, 7, 8, 9, 15, 16, 17, 31, 33, 64, 65, 127, 129, 200, 220, 255, 256, 257, 300]
            Tk = rng.choice(lengths) if rng.random() < 0.8 else rng.randint(300, 1100)
            Tq = Tk if rng.random() < 0.8 else rng.randint(1, Tk)
            q = torch.randn(B, Tq, E, generator=g)
            k, v = torch.randn(B, Tk, E, generator=g), torch.randn(B, Tk, E, generator=g)
            dout = torch.randn(B, Tq, E, generator=g)
            mask = None
            if rng.random() < 0.6:
                mask = torch.rand(B, Tk, generator=g) > 0.3
                mask[:, 0] = True
                if rng.random() < 0.3:
                    mask[-1] = False          # a fully padded sample
            scale = 1.0 / math.sqrt(E)
            qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))
            ref = _ref(qr, kr, vr, mask, heads, scale)
            ref.backward(dout.double())
            qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
            mu8 = ops._mask_u8(mask.cuda()) if mask is not None else None
            out, lse = ops.attention_fwd(qc, kc, vc, mu8, heads, scale)
            what = (B, Tq, Tk, heads, hd, mask is not None, path)
            torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=1e-4, atol=3e-5, msg=lambda m: f"{what}: {m}")
            dq, dk, dv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
            ops.attention_bwd(qc, kc, vc, mu8, heads, scale, out, lse, dout.cuda(), dq, dk, dv)
            for got, want, name in ((dq, qr.grad, "dq"), (dk, kr.grad, "dk"), (dv, vr.grad, "dv")):
                torch.testing.assert_close(got.cpu().double(), want, rtol=3e-4, atol=3e-5,
                                           msg=lambda m: f"{name} {what}: {m}")
    finally:
        _lib.lib().msn_set_attention_path(0)


@pytest.mark.parametrize("T", [17, 33, 49, 65, 81, 97, 113])
@pytest.mark.parametrize("hd,heads", [(64, 6), (32, 2), (16, 2), (8, 4), (48, 1), (24, 3), (40, 2)])
@pytest.mark.parametrize("masked", [False, True])
def test_ragged_last_token_path(T, hd, heads, masked):
    """Self-attention over 16n + 1 tokens (class token + patch grid): the matrix-core kernels run the n x n full tiles and
    the last token goes through their vector-ALU tail (attention_mfma.hip).  Same fp64 reference, incl. a fully padded
    sample and a masked-out last token."""
    from multimodal_supernovae_amd import _lib, ops
    g = torch.Generator().manual_seed(1000 * T + hd)
    B, E = 3, heads * hd
    q, k, v, dout = (torch.randn(B, T, E, generator=g) for _ in range(4))
    mask = None
    if masked:
        mask = torch.rand(B, T, generator=g) > 0.3
        mask[:, 0] = True
        mask[0, -1] = False            # the ragged token itself masked out as a key
        mask[1, -1] = True
        mask[2] = False                # a fully padded sample (uniform attention over the -1e7 scores)
    scale = 1.0 / math.sqrt(E)
    qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))
    ref = _ref(qr, kr, vr, mask, heads, scale)
    ref.backward(dout.double())
    _lib.check(_lib.lib().msn_set_attention_path(2))
    try:
        qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
        mu8 = ops._mask_u8(mask.cuda()) if mask is not None else None
        out, lse = ops.attention_fwd(qc, kc, vc, mu8, heads, scale)
        torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=1e-4, atol=3e-5)
        dq, dk, dv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
        ops.attention_bwd(qc, kc, vc, mu8, heads, scale, out, lse, dout.cuda(), dq, dk, dv)
        for got, want, name in ((dq, qr.grad, "dq"), (dk, kr.grad, "dk"), (dv, vr.grad, "dv")):
            torch.testing.assert_close(got.cpu().double(), want, rtol=3e-4, atol=3e-5, msg=lambda m: f"{name}: {m}")
    finally:
        _lib.lib().msn_set_attention_path(0)


@pytest.mark.parametrize("T", [(9, 9), (65, 65), (128, 128), (40, 128), (129, 129), (200, 200), (77, 300)])
@pytest.mark.parametrize("hd,heads", [(128, 2), (96, 1), (80, 2), (112, 1), (72, 1), (100, 1)])
@pytest.mark.parametrize("masked", [False, True])
def test_wide_heads_on_the_matrix_cores(T, hd, heads, masked):
    """Head widths 65 .. 128 (the reference's default transformer_kwargs: emb 256 / 2 heads = 128-wide heads) take the matrix-core
    kernels too (eight waves of up to 256 registers; a width that is not a multiple of 16 runs as the next one with zero columns),
    short (<= 128 tokens) and chunked (longer) forms, Tq != Tk included; the result must also equal the vector-ALU kernels'."""
    from multimodal_supernovae_amd import _lib, ops
    Tq, Tk = T
    g = torch.Generator().manual_seed(7000 + 10 * Tk + hd)
    B, E = 3, heads * hd
    q = torch.randn(B, Tq, E, generator=g)
    k, v = torch.randn(B, Tk, E, generator=g), torch.randn(B, Tk, E, generator=g)
    dout = torch.randn(B, Tq, E, generator=g)
    mask = None
    if masked:
        mask = torch.rand(B, Tk, generator=g) > 0.3
        mask[:, 0] = True
        mask[2] = False                # a fully padded sample
    scale = 1.0 / math.sqrt(E)
    qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))
    ref = _ref(qr, kr, vr, mask, heads, scale)
    ref.backward(dout.double())
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    mu8 = ops._mask_u8(mask.cuda()) if mask is not None else None
    try:
        for path in (0, 1):            # automatic (= matrix cores for these widths) and the vector-ALU kernels
            _lib.check(_lib.lib().msn_set_attention_path(path))
            out, lse = ops.attention_fwd(qc, kc, vc, mu8, heads, scale)
            torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=1e-4, atol=3e-5, msg=lambda m: f"path {path}: {m}")
            dq, dk, dv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
            ops.attention_bwd(qc, kc, vc, mu8, heads, scale, out, lse, dout.cuda(), dq, dk, dv)
            for got, want, name in ((dq, qr.grad, "dq"), (dk, kr.grad, "dk"), (dv, vr.grad, "dv")):
                torch.testing.assert_close(got.cpu().double(), want, rtol=3e-4, atol=3e-5, msg=lambda m: f"{name} path {path}: {m}")
    finally:
        _lib.lib().msn_set_attention_path(0)


@pytest.mark.parametrize("B,heads,hd,T", [(8, 8, 8, 50), (16, 3, 8, 200), (24, 2, 16, 220), (8, 5, 4, 33), (16, 6, 64, 65), (8, 2, 16, 220), (32, 8, 8, 12)])
def test_xcd_grouped_head_order(B, heads, hd, T):
    """Batches that are multiples of 8 take the XCD-grouped (batch, head) order of the vector-ALU kernels (the heads of a
    sample on one XCD, attention.hip locate_head): every (b, h) must still be computed exactly once."""
    from multimodal_supernovae_amd import _lib, ops
    g = torch.Generator().manual_seed(B * 1000 + T)
    E = heads * hd
    q, k, v, dout = (torch.randn(B, T, E, generator=g) for _ in range(4))
    mask = torch.rand(B, T, generator=g) > 0.3
    mask[:, 0] = True
    scale = 1.0 / math.sqrt(E)
    qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))
    ref = _ref(qr, kr, vr, mask, heads, scale)
    ref.backward(dout.double())
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    mu8 = ops._mask_u8(mask.cuda())
    try:
        for path in (1, 2):            # vector-ALU kernels, matrix-core kernels (same re-ordering in both families)
            _lib.check(_lib.lib().msn_set_attention_path(path))
            out, lse = ops.attention_fwd(qc, kc, vc, mu8, heads, scale)
            torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=1e-4, atol=3e-5)
            dq, dk, dv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
            ops.attention_bwd(qc, kc, vc, mu8, heads, scale, out, lse, dout.cuda(), dq, dk, dv)
            for got, want, name in ((dq, qr.grad, "dq"), (dk, kr.grad, "dk"), (dv, vr.grad, "dv")):
                torch.testing.assert_close(got.cpu().double(), want, rtol=3e-4, atol=3e-5,
                                           msg=lambda m: f"{name} path {path}: {m}")
    finally:
        _lib.lib().msn_set_attention_path(0)


@pytest.mark.parametrize("B,heads,hd,Tq,Tk", [(2, 2, 16, 1024, 1024), (1, 2, 16, 300, 300), (2, 1, 64, 257, 257),
                                               (1, 3, 32, 520, 390), (2, 2, 16, 100, 700), (1, 2, 8, 513, 513)])
def test_long_sequence_matrix_core_path(B, heads, hd, Tq, Tk):
    """More than 256 tokens (the reference's spectrum transformer on 1024-bin spectra): the chunked matrix-core kernels
    (online softmax across key chunks; attention_mfma.hip) against the fp64 reference, incl. a fully padded sample."""
    from multimodal_supernovae_amd import _lib, ops
    g = torch.Generator().manual_seed(Tq * 7 + Tk)
    E = heads * hd
    q = torch.randn(B, Tq, E, generator=g)
    k, v = torch.randn(B, Tk, E, generator=g), torch.randn(B, Tk, E, generator=g)
    dout = torch.randn(B, Tq, E, generator=g)
    mask = torch.rand(B, Tk, generator=g) > 0.3
    mask[:, 0] = True
    mask[-1] = B > 1              # last sample fully padded when there is more than one
    mask[0, 256:300] = False      # a masked stretch across the first chunk boundary
    scale = 1.0 / math.sqrt(E)
    qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))
    ref = _ref(qr, kr, vr, mask, heads, scale)
    ref.backward(dout.double())
    _lib.check(_lib.lib().msn_set_attention_path(2))
    try:
        qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
        mu8 = ops._mask_u8(mask.cuda())
        out, lse = ops.attention_fwd(qc, kc, vc, mu8, heads, scale)
        torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=1e-4, atol=3e-5)
        dq, dk, dv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
        ops.attention_bwd(qc, kc, vc, mu8, heads, scale, out, lse, dout.cuda(), dq, dk, dv)
        for got, want, name in ((dq, qr.grad, "dq"), (dk, kr.grad, "dk"), (dv, vr.grad, "dv")):
            torch.testing.assert_close(got.cpu().double(), want, rtol=3e-4, atol=3e-5, msg=lambda m: f"{name}: {m}")
    finally:
        _lib.lib().msn_set_attention_path(0)


@pytest.mark.parametrize("hd,heads", [(33, 2), (50, 1), (70, 2), (127, 1)])
@pytest.mark.parametrize("T", [40, 200])
def test_wide_heads_of_odd_width_are_padded_on_the_host(hd, heads, T):
    """Heads wider than 32 whose width is not a multiple of 4 (or whose rows are not 16-byte aligned): the binding pads them
    with zero columns and the matrix-core kernels take them (the spilling 64- / 128-wide vector-ALU kernels are gone); the
    C-ABI itself refuses such a width."""
    from multimodal_supernovae_amd import _lib, ops
    g = torch.Generator().manual_seed(hd * 7 + T)
    B, E = 2, heads * hd
    q, k, v, dout = (torch.randn(B, T, E, generator=g) for _ in range(4))
    mask = torch.rand(B, T, generator=g) > 0.3
    mask[:, 0] = True
    scale = 1.0 / math.sqrt(E)
    qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))
    ref = _ref(qr, kr, vr, mask, heads, scale)
    ref.backward(dout.double())
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    mu8 = ops._mask_u8(mask.cuda())
    out, lse = ops.attention_fwd(qc, kc, vc, mu8, heads, scale)
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=1e-4, atol=3e-5)
    dq, dk, dv = (torch.full_like(t, float("nan")) for t in (qc, kc, vc))
    ops.attention_bwd(qc, kc, vc, mu8, heads, scale, out, lse, dout.cuda(), dq, dk, dv)
    for got, want in ((dq, qr.grad), (dk, kr.grad), (dv, vr.grad)):
        torch.testing.assert_close(got.cpu().double(), want, rtol=3e-4, atol=3e-5)
    if hd % 4:
        lse_c = torch.empty(B, heads, T, 2, device="cuda")
        rc = _lib.lib().msn_attention_fwd(ops.ptr(qc), E, T * E, ops.ptr(kc), E, T * E, ops.ptr(vc), E, T * E, ops.ptr(mu8), B, heads, T, T,
                                          hd, scale, ops.ptr(out), E, T * E, ops.ptr(lse_c), ops.stream_ptr())
        assert rc == 1, "the C-ABI must refuse a head wider than 32 that is not a multiple of 4"


@pytest.mark.parametrize("B,T,hd,heads,masked", [(4, 65, 64, 6, False), (3, 65, 64, 6, True), (2, 128, 32, 3, True), (5, 17, 16, 2, False),
                                                 (3, 33, 16, 3, True), (4, 65, 48, 1, False)])
def test_plane_output_backward_against_fp64(B, T, hd, heads, masked):
    """msn_attention_bwd_planes (dqkv leaves the one-pass backward as bf16 planes + column sums) against the fp64 formula --
    until now it was only compared with msn_attention_bwd + msn_plane_split (tests/test_attention_fused_gpu.py)."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(B * 100 + T + hd)
    E = heads * hd
    qkv = torch.randn(B, T, 3 * E, generator=g)
    dout = torch.randn(B, T, E, generator=g)
    mask = None
    if masked:
        mask = torch.rand(B, T, generator=g) > 0.3
        mask[:, 0] = True
        mask[-1] = False
    scale = 1.0 / math.sqrt(hd)
    qr, kr, vr = (qkv[..., i * E:(i + 1) * E].double().requires_grad_() for i in range(3))
    ref = _ref(qr, kr, vr, mask, heads, scale)
    ref.backward(dout.double())
    want = torch.cat([qr.grad, kr.grad, vr.grad], -1).view(B * T, 3 * E)
    qc = qkv.cuda()
    mu8 = ops._mask_u8(mask.cuda()) if masked else None
    out, lse = ops.attention_fwd(qc[..., :E], qc[..., E:2 * E], qc[..., 2 * E:], mu8, heads, scale)
    got, cs = ops.attention_bwd_planes(qc, heads, scale, out, lse, dout.cuda(), 3, want_colsum=True, mask_u8=mu8)
    torch.testing.assert_close(got.to_float().cpu().double(), want, rtol=3e-4, atol=3e-5)
    torch.testing.assert_close(cs.cpu().double(), want.sum(0), rtol=1e-4, atol=1e-4 * float(want.abs().sum(0).max()))


@pytest.mark.parametrize("T", [15, 31, 47, 63])
def test_fused_backward_padded_last_tile_is_stable(T):
    """Sequences of 16 n + 15 tokens: the padding row of the last query tile must not reach the stage of the one-pass backward
    (it is row 0 of the dk block there; written without a barrier it overwrote dk of the first key in some launches)."""
    from multimodal_supernovae_amd import _lib, ops
    g = torch.Generator().manual_seed(T)
    B, heads, hd = 2, 4, 32
    E = heads * hd
    q, k, v, dout = (torch.randn(B, T, E, generator=g) for _ in range(4))
    scale = 1.0 / math.sqrt(E)
    qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))
    _ref(qr, kr, vr, None, heads, scale).backward(dout.double())
    qc, kc, vc, dc = q.cuda(), k.cuda(), v.cuda(), dout.cuda()
    _lib.check(_lib.lib().msn_set_attention_path(2))
    try:
        out, lse = ops.attention_fwd(qc, kc, vc, None, heads, scale)
        first = None
        for _ in range(40):
            dq, dk, dv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
            ops.attention_bwd(qc, kc, vc, None, heads, scale, out, lse, dc, dq, dk, dv)
            got = torch.stack([dq, dk, dv]).cpu()
            if first is None:
                first = got
                for x, want, name in ((dq, qr.grad, "dq"), (dk, kr.grad, "dk"), (dv, vr.grad, "dv")):
                    torch.testing.assert_close(x.cpu().double(), want, rtol=3e-4, atol=3e-5, msg=lambda m: f"{name} T={T}: {m}")
            else:
                assert torch.equal(got, first)
    finally:
        _lib.lib().msn_set_attention_path(0)

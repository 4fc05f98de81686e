"""bf16-resident GEMMs (msn_bgemm_nt / msn_bgemm_tn, gemm_bf16res.hip) and their helpers against torch on the same bf16
inputs in fp64.  Integer-valued operands make every product and partial sum exact in fp32, so the result must be
bit-exact whatever the summation order: that pins the operand / fragment / swizzle maps (an asymmetric B catches a
transposed output); random operands check the epilogues and the accumulation."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ints(shape, g, lo=-4, hi=5):
    return torch.randint(lo, hi, shape, generator=g).to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (16, 8, 64), (300, 264, 192), (1000, 768, 768),
                                   (513, 2304, 768), (777, 768, 3072), (4096, 3072, 768),
                                   (300, 260, 192), (70, 132, 64)])      # N % 8 == 4: bf16 rows not 16-byte aligned (8-byte stores)
def test_nt_exact_on_integers(M, N, K):
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a, w = _ints((M, K), g), _ints((N, K), g)
    bias = torch.randint(-3, 4, (N,), generator=g).float()
    ref = a.double() @ w.double().T + bias.double()
    c = ops.bgemm_nt(a.cuda(), w.cuda(), bias=bias.cuda())
    assert c.dtype == torch.float32 and torch.equal(c.cpu().double(), ref)
    cb = ops.bgemm_nt(a.cuda(), w.cuda(), bias=None, out_bf16=True)
    assert cb.dtype == torch.bfloat16
    torch.testing.assert_close(cb.cpu().double(), (a.double() @ w.double().T).to(torch.bfloat16).double(), rtol=0, atol=0)


@pytest.mark.parametrize("M,N,K", [(300, 264, 192), (2048, 768, 768), (1030, 3072, 768), (300, 260, 192), (513, 388, 128)])
def test_nt_epilogues(M, N, K):
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(7)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    pre_ref = a.double() @ w.double().T + bias.double()
    # ADD: fp32 residual stream
    out = ops.bgemm_nt(a.cuda(), w.cuda(), bias=bias.cuda(), epilogue=ops.BEPI_ADD, aux=res.cuda())
    torch.testing.assert_close(out.cpu().double(), pre_ref + res.double(), rtol=1e-4, atol=1e-4)
    # GELU: bf16 activation + bf16 pre-activation
    f, pre = ops.bgemm_nt(a.cuda(), w.cuda(), bias=bias.cuda(), epilogue=ops.BEPI_GELU, out_bf16=True)
    torch.testing.assert_close(pre.cpu().double(), pre_ref, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(f.cpu().double(), torch.nn.functional.gelu(pre_ref), rtol=1e-2, atol=1e-2)
    # GELU': C = (a . w^T) * gelu'(aux)
    x = pre.float().cpu().double().requires_grad_()
    torch.nn.functional.gelu(x).sum().backward()
    d = ops.bgemm_nt(a.cuda(), w.cuda(), epilogue=ops.BEPI_GELU_BWD, aux=pre, out_bf16=True)
    torch.testing.assert_close(d.cpu().double(), (a.double() @ w.double().T) * x.grad, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("M,N,K", [(64, 256, 256), (100, 264, 520), (5000, 768, 768), (20000, 256, 512), (3152, 2304, 768),
                                   (1, 8, 8)])
def test_tn_exact_on_integers(M, N, K):
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(M + N)
    dy, x = _ints((M, N), g, -2, 3), _ints((M, K), g, -2, 3)
    ref = dy.double().T @ x.double()
    c = ops.bgemm_tn(dy.cuda(), x.cuda())
    assert c.shape == (N, K) and torch.equal(c.cpu().double(), ref)


def test_tn_random_and_strided():
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(3)
    wide = torch.randn(4000, 1024 + 768, generator=g).to(torch.bfloat16).cuda()
    dy, x = wide[:, :1024], wide[:, 1024:]                       # row-strided views (ld = 1792)
    ref = dy.double().T @ x.double()
    c = ops.bgemm_tn(dy, x)
    torch.testing.assert_close(c.double(), ref, rtol=1e-4, atol=1e-3)


def test_casts_colsum_and_layernorm_variants():
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1000, 768, generator=g).cuda()
    assert torch.equal(ops.cast_bf16(x), x.to(torch.bfloat16))
    w = torch.randn(300, 520, generator=g).cuda()
    assert torch.equal(ops.cast_bf16_t(w), w.t().contiguous().to(torch.bfloat16))
    xb = x.to(torch.bfloat16)
    torch.testing.assert_close(ops.bcolsum(xb).double(), xb.double().sum(0), rtol=1e-5, atol=1e-4)
    gamma, beta = torch.randn(768, generator=g).cuda(), torch.randn(768, generator=g).cuda()
    y32, m32, r32 = ops.layernorm_fwd(x, gamma, beta, 1e-6)
    yb, mb, rb = ops.layernorm_fwd_bf16(x, gamma, beta, 1e-6)
    assert torch.equal(yb, y32.to(torch.bfloat16)) and torch.equal(mb, m32) and torch.equal(rb, r32)
    dy, add = torch.randn(1000, 768, generator=g).cuda(), torch.randn(1000, 768, generator=g).cuda()
    dx32, dg32, db32 = ops.layernorm_bwd(dy, x, m32, r32, gamma, add=add)
    dx, dxb, dg, db = ops.layernorm_bwd_bf16(dy, x, m32, r32, gamma, add=add)
    assert torch.equal(dx, dx32) and torch.equal(dxb, dx32.to(torch.bfloat16)) and torch.equal(dg, dg32) and torch.equal(db, db32)
    dx2, dxb2, dg2, db2, cs = ops.layernorm_bwd_bf16(dy, x, m32, r32, gamma, add=add, want_colsum=True)
    assert torch.equal(dx2, dx32) and torch.equal(dg2, dg32) and torch.equal(db2, db32)
    torch.testing.assert_close(cs.double(), dx32.double().sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(300, 264, 192), (4000, 3072, 768), (257, 768, 64), (300, 260, 192)])
def test_nt_epilogue_column_sums(M, N, K):
    """Bias gradients from the epilogue of the product that writes the gradient matrix: out[n] = sum_m C[m][n] of the
    values AS STORED (bf16-rounded when C is bf16), for the plain and the GELU' epilogue."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(M)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    pre = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    c, cs = ops.bgemm_nt(a, w, want_colsum=True)
    assert torch.equal(c, ops.bgemm_nt(a, w))
    torch.testing.assert_close(cs.double(), c.double().sum(0), rtol=1e-5, atol=1e-3)
    d, ds = ops.bgemm_nt(a, w, epilogue=ops.BEPI_GELU_BWD, aux=pre, out_bf16=True, want_colsum=True)
    assert torch.equal(d, ops.bgemm_nt(a, w, epilogue=ops.BEPI_GELU_BWD, aux=pre, out_bf16=True))
    torch.testing.assert_close(ds.double(), d.double().sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("heads", [2, 4])          # 64-wide heads: bf16 attention kernels; 32-wide: fp32 attention + casts
def test_resident_trunk_equals_the_fp32_storage_bf16_path(heads):
    """ViT blocks through functional._Bf16VitTrunk == the same blocks with gemm_precision "bf16" on fp32-stored
    activations, up to bf16 rounding noise (direction of every gradient, tight bound on the output)."""
    from multimodal_supernovae_amd.encoders import VisionTransformer
    torch.manual_seed(9)
    m = VisionTransformer(img_size=32, patch_size=8, emb=128, depth=3, heads=heads, n_out=8, gemm_precision="bf16").cuda()
    x = torch.rand(6, 3, 32, 32, device="cuda")
    cot = torch.randn(6, 8, device="cuda")
    res = []
    for resident in (True, False):
        m.bf16_resident = resident
        m.zero_grad(set_to_none=True)
        y = m(x)
        y.backward(cot)
        res.append((y.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    (ya, ga), (yb, gb) = res
    cos = lambda a, b: float((a.flatten().double() @ b.flatten().double()) / (a.double().norm() * b.double().norm() + 1e-300))
    assert cos(ya, yb) > 0.9995
    low = {k: cos(ga[k], gb[k]) for k in ga if cos(ga[k], gb[k]) < 0.99}
    assert not low, low


@pytest.mark.parametrize("B,H,T", [(2, 2, 16), (3, 2, 65), (2, 12, 197), (1, 1, 256), (2, 3, 33), (1, 2, 1)])
def test_bf16_attention_forward_backward(B, H, T):
    """msn_attention_bf16_fwd / _bwd against softmax attention evaluated in fp64 on the same bf16 q, k, v, dO (the
    kernels round P and dS to bf16 before the second products: tolerance = bf16 resolution of values of that size)."""
    from multimodal_supernovae_amd import ops
    e = 64 * H
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = (torch.randn(B * T, 3 * e, generator=g) * 0.8).to(torch.bfloat16)
    dout = torch.randn(B * T, e, generator=g).to(torch.bfloat16)
    scale = 0.125
    x = qkv.double().view(B, T, 3, H, 64).requires_grad_()
    q, k, v = x[:, :, 0], x[:, :, 1], x[:, :, 2]
    att = torch.softmax(torch.einsum("bihd,bjhd->bhij", q, k) * scale, dim=-1)
    ref = torch.einsum("bhij,bjhd->bihd", att, v).reshape(B * T, e)
    lse_ref = torch.logsumexp(torch.einsum("bihd,bjhd->bhij", q, k) * scale, dim=-1)
    ref.backward(dout.double())
    dref = x.grad.reshape(B * T, 3 * e)
    out, lse = ops.attention_bf16_fwd(qkv.cuda(), B, T, H, scale)
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(lse.cpu().double(), lse_ref.detach(), rtol=1e-4, atol=1e-4)
    dqkv = ops.attention_bf16_bwd(qkv.cuda(), out, dout.cuda(), lse, B, T, H, scale)
    dqkv2, cs = ops.attention_bf16_bwd(qkv.cuda(), out, dout.cuda(), lse, B, T, H, scale, want_colsum=True)
    assert torch.equal(dqkv, dqkv2)                          # bias gradient of the packed projection from inside the passes
    torch.testing.assert_close(cs.double(), dqkv.double().sum(0), rtol=1e-4, atol=1e-3)
    err = (dqkv.cpu().double() - dref).abs().max() / dref.abs().max()
    assert err < 2e-2, float(err)
    cosv = float((dqkv.cpu().double().flatten() @ dref.flatten()) / (dqkv.cpu().double().norm() * dref.norm()))
    assert cosv > 0.9995, cosv


@pytest.mark.parametrize("M,N,K,epi", [(100864, 2304, 768, "none"), (100864, 768, 768, "add"), (100864, 3072, 768, "gelu"), (65700, 768, 3072, "none"),
                                       (70000, 1000, 128, "none")])
def test_nt_persistent_workgroups_equal_one_tile_launches(M, N, K, epi):
    """BASELINE cfg5 sizes (batch 512: 100 864 token rows): the persistent launch (one workgroup per CU walking its tiles, the
    LDS ring running on across tile boundaries) gives the bits of one-workgroup-per-tile launches -- what a product of at most 256
    tiles gets: the same product issued in row slabs of at most 256 tiles each -- ragged last row tile, ragged last column tile,
    epilogues with aux reads / writes."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g) if epi == "add" else None

    def run(rows):
        lo, hi = rows
        if epi == "gelu":
            aux = torch.empty(hi - lo, N, device="cuda", dtype=torch.bfloat16)
            y = ops.bgemm_nt(a[lo:hi], w, bias=bias, epilogue=ops.BEPI_GELU, aux=aux, out_bf16=True)
            return [y, aux]
        if epi == "add":
            return [ops.bgemm_nt(a[lo:hi], w, bias=bias, epilogue=ops.BEPI_ADD, aux=res[lo:hi])]
        return [ops.bgemm_nt(a[lo:hi], w, bias=bias, out_bf16=True)]

    def flat(o):
        return [o] if torch.is_tensor(o) else [t for e in o for t in flat(e)]

    whole = flat(run((0, M)))
    slab = 256 * max(1, 256 // ((N + 255) // 256))          # rows of a slab of at most 256 tiles (whole 256-row tiles)
    parts = [flat(run((lo, min(M, lo + slab)))) for lo in range(0, M, slab)]
    assert len(parts) > 1
    for k, t in enumerate(whole):
        assert torch.equal(t, torch.cat([p[k] for p in parts]))


def test_cast_bf16_list_matches_torch():
    """msn_cast_bf16_list: plain and transposed bf16 copies of a list of matrices from one launch == torch's round-to-nearest-even."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(3)
    mats = [torch.randn(r, c, generator=g).cuda() for r, c in [(2304, 768), (768, 768), (40, 72), (1, 8), (33, 31), (3072, 768)]]
    for transposed in (False, True):
        outs = ops.cast_bf16_list(mats, transposed=transposed)
        for w, y in zip(mats, outs):
            want = (w.t().contiguous() if transposed else w).to(torch.bfloat16)
            assert y.shape == want.shape and torch.equal(y, want)
    assert ops.cast_bf16_list([]) == []

"""fp32-grade GEMMs from resident bf16 planes (msn_pgemm_nt / msn_pgemm_tn / msn_plane_split, csrc/pgemm.hip) against
torch in fp64.

* three planes hold an fp32 value exactly: split -> merge is the identity, bit for bit;
* integer-valued operands make every product and partial sum exact in fp32, so the result must be bit-exact whatever the
  summation order: that pins the block layout, the LDS-DMA / fragment / swizzle maps and the tile edges (an asymmetric B
  catches a transposed output);
* THE GATE for using the 3-plane form where the product promises fp32: on every GEMM shape of the
  headline step, on N(0,1), cancellation-heavy and wide-dynamic-range operands, maximum and RMS error against fp64 at
  most 1.5 x those of the native fp32 MFMA kernel (msn_sgemm)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ints(shape, g, lo=-4, hi=5):
    return torch.randint(lo, hi, shape, generator=g).float()


@pytest.mark.parametrize("R,C", [(32, 32), (100, 70), (257, 384), (1000, 1152), (31, 17)])
@pytest.mark.parametrize("planes", [3, 2])
def test_split_merge(R, C, planes):
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(R * 7 + C)
    x = torch.randn(R, C, generator=g) * torch.exp(torch.randn(R, C, generator=g) * 4)      # wide dynamic range
    xp, cs = ops.plane_split(x.cuda(), planes=planes, want_colsum=True)
    y = xp.to_float().cpu()
    if planes == 3:
        assert torch.equal(y, x), "three bf16 planes must hold an fp32 value exactly"
    else:
        assert float(((y - x).abs() / x.abs().clamp_min(1e-30)).max()) < 2.0 ** -15
    torch.testing.assert_close(cs.cpu().double(), x.double().sum(0), rtol=1e-5, atol=1e-5 * float(x.abs().sum(0).max()))
    xt = ops.plane_split(x.cuda(), planes=planes, transposed=True)
    assert (xt.R, xt.C) == (C, R)
    yt = xt.to_float().cpu()
    if planes == 3:
        assert torch.equal(yt, x.T.contiguous())
    # strided rows (a column slice of a wider matrix)
    wide = torch.randn(R, C + 8, generator=g).cuda()
    assert torch.equal(ops.plane_split(wide[:, 4:4 + C], planes=3).to_float(), wide[:, 4:4 + C])


@pytest.mark.parametrize("M,N,K", [(256, 256, 32), (256, 128, 64), (300, 272, 200), (1000, 384, 384), (513, 1152, 384),
                                   (777, 384, 1536), (4096, 1536, 384), (260, 144, 132), (65 * 40, 384, 192)])
@pytest.mark.parametrize("planes", [3, 2])
def test_nt_exact_on_integers(M, N, K, planes):
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a, w = _ints((M, K), g), _ints((N, K), g)
    bias = torch.randint(-3, 4, (N,), generator=g).float()
    ref = a.double() @ w.double().T + bias.double()
    ap, wp = ops.plane_split(a.cuda(), planes), ops.plane_split(w.cuda(), planes)
    c = ops.pgemm_nt(ap, wp, bias=bias.cuda())
    assert c.dtype == torch.float32 and torch.equal(c.cpu().double(), ref)
    # the operand of the dgrad products: planes of the transposed weight
    wt = ops.plane_split(w.T.contiguous().cuda(), planes, transposed=True)
    assert torch.equal(ops.pgemm_nt(ap, wt).cpu().double(), a.double() @ w.double().T)
    if N % 16 == 0:
        junk = ops.Planes.empty(M, N, planes, "cuda")   # poison what the allocator hands out next: N % 32 == 16 has a padding
        junk.buf.fill_(0x7F)                             # column block that the kernel must write (zeros), not leave as found
        del junk
        cp, cs = ops.pgemm_nt(ap, wp, bias=bias.cuda(), out_planes=True, want_colsum=True)
        assert torch.equal(cp.to_float().cpu().double(), ref)
        assert torch.equal(cp.buf, ops.plane_split(ref.float().cuda(), planes).buf), "padding rows / columns of a plane output"
        assert torch.equal(cs.cpu().double(), ref.sum(0))
        # a plane output is an operand: rows / columns of its padding must be zero (they enter the TN reduction)
        z = ops.pgemm_tn(cp, ap)
        assert torch.equal(z.cpu().double(), ref.T @ a.double())


@pytest.mark.parametrize("M,N,K", [(300, 272, 200), (2048, 384, 384), (1030, 1536, 384)])
def test_nt_epilogues(M, N, K):
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(7)
    a = torch.randn(M, K, generator=g) * 0.5
    w = torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    pre_ref = a.double() @ w.double().T + bias.double()
    ap, wp = ops.plane_split(a.cuda(), 3), ops.plane_split(w.cuda(), 3)
    out = ops.pgemm_nt(ap, wp, bias=bias.cuda(), epilogue=ops.EPI_ADD, aux=res.cuda())
    torch.testing.assert_close(out.cpu().double(), pre_ref + res.double(), rtol=1e-5, atol=1e-5)
    f, dact = ops.pgemm_nt(ap, wp, bias=bias.cuda(), epilogue=ops.EPI_GELU, aux=True, out_planes=True)
    torch.testing.assert_close(f.to_float().cpu().double(), torch.nn.functional.gelu(pre_ref), rtol=1e-5, atol=1e-5)
    x = pre_ref.clone().requires_grad_()
    torch.nn.functional.gelu(x).sum().backward()
    torch.testing.assert_close(dact.cpu().double(), x.grad, rtol=1e-5, atol=1e-5)
    d, cs = ops.pgemm_nt(ap, wp, epilogue=ops.EPI_GELU_BWD, aux=dact, out_planes=True, want_colsum=True)
    dref = (a.double() @ w.double().T) * dact.cpu().double()
    torch.testing.assert_close(d.to_float().cpu().double(), dref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(cs.cpu().double(), dref.sum(0), rtol=1e-4, atol=1e-4)
    r = ops.pgemm_nt(ap, wp, bias=bias.cuda(), epilogue=ops.EPI_RELU)
    torch.testing.assert_close(r.cpu().double(), pre_ref.clamp_min(0), rtol=1e-5, atol=1e-5)
    rb = ops.pgemm_nt(ap, wp, epilogue=ops.EPI_RELU_BWD, aux=r)
    torch.testing.assert_close(rb.cpu().double(), (a.double() @ w.double().T) * (r.cpu() > 0), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M,N,K", [(64, 256, 256), (100, 264, 520), (5000, 384, 384), (20000, 1152, 384), (3152, 384, 1536),
                                   (6656, 1536, 384), (33, 16, 8), (1, 128, 128)])
@pytest.mark.parametrize("planes", [3, 2])
def test_tn_exact_on_integers(M, N, K, planes):
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(M + N)
    dy, x = _ints((M, N), g, -2, 3), _ints((M, K), g, -2, 3)
    ref = dy.double().T @ x.double()
    c = ops.pgemm_tn(ops.plane_split(dy.cuda(), planes), ops.plane_split(x.cuda(), planes))
    assert torch.equal(c.cpu().double(), ref)


def _operands(kind, M, N, K, g):
    if kind == "normal":
        return torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    if kind == "cancel":       # every inner product is a sum of large terms that cancel to O(1)
        a = torch.randn(M, K, generator=g) * 100
        a[:, 1::2] = -a[:, 0::2] + torch.randn(M, K // 2, generator=g) * 0.01
        w = torch.randn(N, K, generator=g)
        w[:, 1::2] = w[:, 0::2]
        return a, w
    # wide dynamic range: exponents spread over 2^+-20, per element
    a = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-20, 21, (M, K), generator=g).float())
    w = torch.randn(N, K, generator=g) * torch.exp2(torch.randint(-20, 21, (N, K), generator=g).float())
    return a, w


HEADLINE_SHAPES = [   # (N, K) of the ViT-S/8 block products, forward and input-gradient (rows M = tokens)
    (1152, 384), (384, 384), (1536, 384), (384, 1536), (384, 1152), (384, 192)]


def _planes_code(arith):
    from multimodal_supernovae_amd import ops
    return {"bf16x6": 3, "f16x3": ops.F16_PLANES}[arith]


@pytest.mark.parametrize("N,K", HEADLINE_SHAPES)
@pytest.mark.parametrize("kind", ["normal", "cancel", "wide", "tiny", "huge"])
@pytest.mark.parametrize("arith", ["bf16x6", "f16x3"])
def test_fp32_grade_gate_nt(N, K, kind, arith):
    """max and RMS error of the plane GEMMs against fp64 <= 1.5 x the native fp32 MFMA kernel's: three bf16 planes / six
    products (the default arithmetic: passes everywhere), and two fp16 planes / three products with the per-matrix
    power-of-two scale (tiny / huge: the whole matrix outside fp16's own range).  The fp16 form passes on every kind of
    data but one: its operands carry 22 significand bits, not 24, and where the inner products CANCEL (native: exact
    products, the error is the last rounding) that shows -- measured 2.2 - 2.7 x the native kernel's.  That is why it is an
    opt-in of the C-ABI and not the default; the bound asserted for it there is 3 x."""
    from multimodal_supernovae_amd import ops
    M = 2080                                             # 32 cutouts x 65 tokens: the error statistics do not depend on M
    g = torch.Generator().manual_seed(N + K)
    a, w = _operands("normal" if kind in ("tiny", "huge") else kind, M, N, K, g)
    if kind == "tiny":
        a, w = a * 1e-12, w * 3e-9
    if kind == "huge":
        a, w = a * 1e9, w * 7e5
    pc = _planes_code(arith)
    ref = a.double() @ w.double().T
    c_nat = ops.sgemm(a.cuda(), w.cuda(), ops.OP_N, ops.OP_T, precision=ops.PREC_F32).cpu().double()
    c_pl = ops.pgemm_nt(ops.plane_split(a.cuda(), pc), ops.plane_split(w.cuda(), pc)).cpu().double()
    e_nat, e_pl = (c_nat - ref).abs(), (c_pl - ref).abs()
    lim = 3.0 if (arith == "f16x3" and kind == "cancel") else 1.5
    assert float(e_pl.max()) <= lim * float(e_nat.max()) + 1e-30, (float(e_pl.max()), float(e_nat.max()))
    assert float(e_pl.pow(2).mean().sqrt()) <= lim * float(e_nat.pow(2).mean().sqrt()) + 1e-30


@pytest.mark.parametrize("N,K", [(1152, 384), (384, 384), (1536, 384), (384, 1536)])
@pytest.mark.parametrize("kind", ["normal", "cancel", "wide"])
@pytest.mark.parametrize("arith", ["bf16x6", "f16x3"])
def test_fp32_grade_gate_tn(N, K, kind, arith):
    from multimodal_supernovae_amd import ops
    pc = _planes_code(arith)
    M = 66560 // 8
    g = torch.Generator().manual_seed(N * 3 + K)
    # the reduction runs over the rows here: build the operands reduction-major
    at, wt = _operands(kind, N, K, M, g)                 # (N, M), (K, M)
    dy, x = at.T.contiguous(), wt.T.contiguous()
    ref = dy.double().T @ x.double()
    c_nat = ops.sgemm(dy.cuda(), x.cuda(), ops.OP_T, ops.OP_N, precision=ops.PREC_F32).cpu().double()
    c_pl = ops.pgemm_tn(ops.plane_split(dy.cuda(), pc), ops.plane_split(x.cuda(), pc)).cpu().double()
    e_nat, e_pl = (c_nat - ref).abs(), (c_pl - ref).abs()
    lim = 3.0 if (arith == "f16x3" and kind == "cancel") else 1.5
    assert float(e_pl.max()) <= lim * float(e_nat.max()) + 1e-30, (float(e_pl.max()), float(e_nat.max()))
    assert float(e_pl.pow(2).mean().sqrt()) <= lim * float(e_nat.pow(2).mean().sqrt()) + 1e-30


def _report(line):
    """Measured gate ratios -> gpurun_out/r06_pgemm_accuracy.txt (copied to profiles/ by hand)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "r06_pgemm_accuracy.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def _operands_cuda(kind, M, N, K, seed):
    """_operands on the device (the full-length weight-gradient gate builds 66 560-row operands)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
    if kind == "normal":
        return rn(M, K), rn(N, K)
    if kind == "positive":     # every term of every inner product has one sign: truncation errors cannot average out
        return rn(M, K).abs(), rn(N, K).abs()
    if kind == "cancel":
        a = rn(M, K) * 100
        a[:, 1::2] = -a[:, 0::2] + rn(M, K // 2) * 0.01
        w = rn(N, K)
        w[:, 1::2] = w[:, 0::2]
        return a, w
    ea = torch.randint(-20, 21, (M, K), generator=g, device="cuda").float()
    ew = torch.randint(-20, 21, (N, K), generator=g, device="cuda").float()
    return rn(M, K) * torch.exp2(ea), rn(N, K) * torch.exp2(ew)


@pytest.mark.parametrize("N,K", [(1152, 384), (384, 384), (1536, 384), (384, 1536)])
@pytest.mark.parametrize("kind", ["normal", "cancel", "wide", "positive"])
def test_fp32_grade_gate_tn_full_length(N, K, kind):
    """The weight-gradient gate at the headline's REAL reduction length (66 560 token rows; the test above reduces over an
    eighth of it): max and RMS error against fp64 <= 1.5 x the native fp32 kernel's, default arithmetic.  "positive" is the
    case the kernel's two-step temporaries (FOLDN = 2) could lose: one-signed terms, every truncation biased the same way."""
    from multimodal_supernovae_amd import ops
    M = 66560
    at, wt = _operands_cuda(kind, N, K, M, N * 5 + K)     # (N, M), (K, M): reduction-major
    dy, x = at.T.contiguous(), wt.T.contiguous()
    del at, wt
    ref = dy.double().T @ x.double()
    c_nat = ops.sgemm(dy, x, ops.OP_T, ops.OP_N, precision=ops.PREC_F32).double()
    c_pl = ops.pgemm_tn(ops.plane_split(dy, 3), ops.plane_split(x, 3)).double()
    e_nat, e_pl = (c_nat - ref).abs(), (c_pl - ref).abs()
    rmax = float(e_pl.max()) / (float(e_nat.max()) + 1e-300)
    rrms = float(e_pl.pow(2).mean().sqrt()) / (float(e_nat.pow(2).mean().sqrt()) + 1e-300)
    _report(f"TN full length M=66560 N={N} K={K} {kind}: plane / native error  max {rmax:.3f}  rms {rrms:.3f}")
    assert rmax <= 1.5 and rrms <= 1.5, (rmax, rrms)


def test_fp32_grade_gate_step_operands():
    """The gate on the ACTUAL operands of a headline step instead of synthetic ones: every plane product of one
    forward + backward of the ViT-S/8 tower at 1024 cutouts is intercepted (LayerNorm-output planes x weights, the GELU
    activation, GELU'-scaled gradients, dqkv from the attention backward, 66 560-row weight-gradient reductions); for the
    first product of each distinct (kind, shape, operand role) the bare product is recomputed three ways -- fp64 (torch),
    the native fp32 MFMA kernel, the plane kernel -- from the very planes the step multiplied, and the same 1.5 x rule on
    maximum and RMS error is applied."""
    from multimodal_supernovae_amd import ops, encoders
    torch.manual_seed(3)
    enc = encoders.vit_s8().cuda().train()
    img = torch.rand(1024, 3, 64, 64, device="cuda")
    seen, rows = set(), []
    real_nt, real_tn = ops.pgemm_nt, ops.pgemm_tn

    def measure(tag, a, w, tn):
        key = (tag, a.R, a.C, w.R, w.C)
        if key in seen or len([k for k in seen if k[0] == tag]) >= 8:
            return
        seen.add(key)
        af, wf = a.to_float(), w.to_float()
        if tn:
            ref = af.double().T @ wf.double()
            nat = ops.sgemm(af, wf, ops.OP_T, ops.OP_N, precision=ops.PREC_F32).double()
            pl = real_tn(a, w).double()
        else:
            ref = af.double() @ wf.double().T
            nat = ops.sgemm(af, wf, ops.OP_N, ops.OP_T, precision=ops.PREC_F32).double()
            pl = real_nt(a, w).double()
        e_nat, e_pl = (nat - ref).abs(), (pl - ref).abs()
        rows.append((key, float(e_pl.max()) / (float(e_nat.max()) + 1e-300),
                     float(e_pl.pow(2).mean().sqrt()) / (float(e_nat.pow(2).mean().sqrt()) + 1e-300)))

    def nt(a, w, *args, **kw):
        measure("NT", a, w, False)
        return real_nt(a, w, *args, **kw)

    def tn(dy, x):
        measure("TN", dy, x, True)
        return real_tn(dy, x)

    ops.pgemm_nt, ops.pgemm_tn = nt, tn
    try:
        out = enc(img)
        (out * torch.randn_like(out)).sum().backward()
        torch.cuda.synchronize()
    finally:
        ops.pgemm_nt, ops.pgemm_tn = real_nt, real_tn
    assert len(rows) >= 8, "the plane path did not run (GEMM precision?)"
    for key, rmax, rrms in rows:
        _report(f"step operands {key[0]} A {key[1]}x{key[2]} B {key[3]}x{key[4]}: plane / native error  max {rmax:.3f}  rms {rrms:.3f}")
    bad = [r for r in rows if r[1] > 1.5 or r[2] > 1.5]
    assert not bad, bad


@pytest.mark.parametrize("M,N,K", [(66560, 384, 768), (33280, 384, 1536), (20000, 1152, 1152), (66560, 384, 1152),
                                   (8320, 384, 1536), (8320, 384, 384), (16640, 384, 768), (8000, 1152, 384)])
def test_nt_tail_split(M, N, K):
    """Tiles that do not fill a round of the 256 persistent workgroups are cut into K-segments + a finishing launch: exact on
    integers, and equal to the unsplit launch up to the summation order on random data; epilogues NONE / ADD / RELU."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(M + N)
    a, w = _ints((M, K), g, -2, 3).cuda(), _ints((N, K), g, -2, 3).cuda()
    bias = torch.randint(-3, 4, (N,), generator=g).float().cuda()
    res = torch.randint(-3, 4, (M, N), generator=g).float().cuda()
    ap, wp = ops.plane_split(a, 3), ops.plane_split(w, 3)
    ref = a.double() @ w.double().T + bias.double()
    assert torch.equal(ops.pgemm_nt(ap, wp, bias=bias).double(), ref)
    assert torch.equal(ops.pgemm_nt(ap, wp, bias=bias, epilogue=ops.EPI_ADD, aux=res).double(), ref + res.double())
    assert torch.equal(ops.pgemm_nt(ap, wp, bias=bias, epilogue=ops.EPI_RELU).double(), ref.clamp_min(0))
    x, y = torch.randn(M, K, generator=g).cuda(), torch.randn(N, K, generator=g).cuda()
    xp, yp = ops.plane_split(x, 3), ops.plane_split(y, 3)
    c1 = ops.pgemm_nt(xp, yp)
    ops.set_pgemm_tail_split(False)
    try:
        c0 = ops.pgemm_nt(xp, yp)
    finally:
        ops.set_pgemm_tail_split(True)
    torch.testing.assert_close(c1, c0, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(66560, 384, 384), (66560, 1536, 384), (40000, 384, 1536)])
def test_full_scale_exact_on_integers(M, N, K):
    """The headline token count: persistent workgroups walk several tiles each, the ring of K-steps runs on across tile
    boundaries, long reductions go in K chunks, tail tiles are cut.  Every output form, twice (two faults of this kind were
    intermittent: a copy of a fragment register taken before its LDS read had landed, and a VALU write to the data registers
    of a 16-byte LDS store one wait state early -- both only with several tiles per workgroup)."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(M + K)
    a, w = _ints((M, K), g, -2, 3).cuda(), _ints((N, K), g, -2, 3).cuda()
    bias = torch.randint(-3, 4, (N,), generator=g).float().cuda()
    res = torch.randint(-3, 4, (M, N), generator=g).float().cuda()
    dact = torch.randint(-3, 4, (M, N), generator=g).float().cuda()
    ap, wp = ops.plane_split(a, 3), ops.plane_split(w, 3)
    prod = a.double() @ w.double().T
    for _ in range(2):
        assert torch.equal(ops.pgemm_nt(ap, wp, bias=bias, out_planes=True).to_float().double(), prod + bias.double())
        cp, cs = ops.pgemm_nt(ap, wp, epilogue=ops.EPI_GELU_BWD, aux=dact, out_planes=True, want_colsum=True)
        assert torch.equal(cp.to_float().double(), prod * dact.double()) and torch.equal(cs.double(), (prod * dact.double()).sum(0))
        assert torch.equal(ops.pgemm_nt(ap, wp, bias=bias, epilogue=ops.EPI_ADD, aux=res).double(), prod + bias.double() + res.double())
        assert torch.equal(ops.pgemm_nt(ap, wp).double(), prod)
        assert torch.equal(ops.pgemm_tn(ap, ap).double(), a.double().T @ a.double())


def test_deterministic():
    """Same inputs, same bits: the tail split's finishing launch, the K chunks, the TN slabs and the column sums all add in a
    fixed order (no floating-point atomics anywhere)."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(9)
    M, N, K = 33280, 384, 1536
    a, w = torch.randn(M, K, generator=g).cuda(), torch.randn(N, K, generator=g).cuda()
    dact = torch.randn(M, N, generator=g).cuda()
    ap, wp = ops.plane_split(a, 3), ops.plane_split(w, 3)
    r1 = (ops.pgemm_nt(ap, wp), ops.pgemm_tn(ap, ap), *ops.pgemm_nt(ap, wp, epilogue=ops.EPI_GELU_BWD, aux=dact, out_planes=True, want_colsum=True))
    r2 = (ops.pgemm_nt(ap, wp), ops.pgemm_tn(ap, ap), *ops.pgemm_nt(ap, wp, epilogue=ops.EPI_GELU_BWD, aux=dact, out_planes=True, want_colsum=True))
    for x, y in zip(r1, r2):
        x, y = (x.buf, y.buf) if isinstance(x, ops.Planes) else (x, y)
        assert torch.equal(x, y)


@pytest.mark.parametrize("planes", [3, 2])
@pytest.mark.parametrize("transposed", [False, True])
def test_split_list_equals_single_splits(planes, transposed):
    """msn_plane_split_list (several matrices, one launch; more than one table's worth of items) == msn_plane_split of each."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(11)
    shapes = [(1152, 384), (384, 384), (1536, 384), (384, 1536), (33, 17), (1, 5), (100, 260), (64, 64)] * 9      # 72 items
    mats = [torch.randn(r, c, generator=g).cuda() for r, c in shapes]
    mats[3] = torch.randn(384, 2000, generator=g).cuda()[:, 100:1636]           # a row stride that is not the width
    got = ops.plane_split_list(mats, planes, transposed=transposed)
    for m, p in zip(mats, got):
        want = ops.plane_split(m, planes, transposed=transposed)
        assert (p.R, p.C) == (want.R, want.C) and torch.equal(p.buf, want.buf)


@pytest.mark.parametrize("M,N,K", [(300, 272, 200), (4096, 1536, 384), (777, 384, 1536), (66560 // 4, 384, 384), (33, 16, 8)])
def test_f16_planes_exact_on_integers(M, N, K):
    """Two fp16 planes with their scales: small integers are exact whatever power of two the matrices are scaled by; the
    transposed split shares the scale; bias, epilogues and column sums go through the same epilogue as the bf16 form."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a, w = _ints((M, K), g), _ints((N, K), g)
    bias = _ints((N,), g).cuda()
    for sa, sw in ((1.0, 1.0), (2.0 ** -30, 2.0 ** 9), (2.0 ** 40, 2.0 ** -3)):
        A, W = (a * sa).cuda(), (w * sw).cuda()
        ap, wp = ops.plane_split(A, ops.F16_PLANES), ops.plane_split(W, ops.F16_PLANES)
        ref = (a.double() @ w.double().T) * (sa * sw)
        c, cs = ops.pgemm_nt(ap, wp, want_colsum=True)
        assert torch.equal(c.cpu().double(), ref)
        torch.testing.assert_close(cs.cpu().double(), ref.sum(0), rtol=1e-6, atol=0)
        if sa == 1.0:
            r = ops.pgemm_nt(ap, wp, bias=bias, epilogue=ops.EPI_RELU)
            assert torch.equal(r.cpu().double(), (ref + bias.cpu().double()).clamp_min(0))
        # dX = dY . W through the transposed planes of W (scale shared), dW = dY^T . X
        dy = _ints((M, N), g)
        dyp = ops.plane_split(dy.cuda(), ops.F16_PLANES)
        wt = ops.plane_split(W, ops.F16_PLANES, transposed=True, scale_of=wp)
        assert torch.equal(ops.pgemm_nt(dyp, wt).cpu().double(), (dy.double() @ w.double()) * sw)
        assert torch.equal(ops.pgemm_tn(dyp, ap).cpu().double(), (dy.double().T @ a.double()) * sa)


@pytest.mark.parametrize("M,N,K", [(777, 384, 1536), (2080, 384, 1152), (300, 272, 200)])
def test_column_sums_of_a_chunked_reduction(M, N, K):
    """fp32 output + column sums with a reduction long enough to be cut into K chunks (the partial sums meet in C; the
    sums are taken from the finished C)."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(M + K)
    a, w = _ints((M, K), g), _ints((N, K), g)
    ref = a.double() @ w.double().T
    for pc in (3, ops.F16_PLANES):
        c, cs = ops.pgemm_nt(ops.plane_split(a.cuda(), pc), ops.plane_split(w.cuda(), pc), want_colsum=True)
        assert torch.equal(c.cpu().double(), ref)
        torch.testing.assert_close(cs.cpu().double(), ref.sum(0), rtol=1e-6, atol=0)

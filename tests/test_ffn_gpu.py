"""The fused feed-forward kernels of the narrow towers (csrc/ffn_planes.hip: msn_ffn_fwd / msn_ffn_bwd; emb 32, hidden 128 -- the
feed-forward half of ref src/transformer_utils.py:102-116 for the spectrum transformer) against fp64 torch:

* integer-valued operands make every product and partial sum exact in fp32, so forward AND backward must be bit-exact whatever the
  summation order: that pins the plane layout, the fragment / transposed-read maps, the hidden-chunk walk and the partial sums;
* THE GATE for fp32-grade arithmetic on the bf16 matrix cores: maximum and RMS error against fp64 at most 1.5 x
  those of the native fp32 MFMA path (msn_sgemm with its ReLU / ReLU' / residual epilogues) on the same operands;
* determinism (fixed-order partial sums), ragged row counts, rows that are slices of a wider matrix."""
import pytest
import torch

pytestmark = pytest.mark.gpu
E, HID = 32, 128


def _ref(x, w1, c1, w2, c2, dz):
    x = x.double().requires_grad_()
    w1, c1, w2, c2 = (t.double().requires_grad_() for t in (w1, c1, w2, c2))
    z = torch.relu(x @ w1.T + c1) @ w2.T + c2 + x
    z.backward(dz.double())
    return z.detach(), x.grad, w1.grad, c1.grad, w2.grad, c2.grad


def _fused(x, w1, c1, w2, c2, dz):
    from multimodal_supernovae_amd import ops
    w1p, w2tp = ops.ffn_weight_planes(w1, w2)
    z = ops.ffn_fwd(x, w1p, w2tp, c1, c2)
    return (z,) + tuple(ops.ffn_bwd(x, dz, w1p, w2tp, c1))


def _native(x, w1, c1, w2, c2, dz):
    """The unfused fp32 path of functional._PostNormBlock (native fp32 MFMA kernels)."""
    from multimodal_supernovae_amd import ops
    P = ops.PREC_F32
    hdn = ops.sgemm(x, w1, ops.OP_N, ops.OP_T, bias=c1, epilogue=ops.EPI_RELU, precision=P)
    z = ops.sgemm(hdn, w2, ops.OP_N, ops.OP_T, bias=c2, epilogue=ops.EPI_ADD, aux=x, precision=P)
    dw2, dc2 = ops.wgrad_bias(dz, hdn, precision=P)
    dpre = ops.sgemm(dz, w2, ops.OP_N, ops.OP_N, epilogue=ops.EPI_RELU_BWD, aux=hdn, precision=P)
    dw1, dc1 = ops.wgrad_bias(dpre, x, precision=P)
    dx = ops.sgemm(dpre, w1, ops.OP_N, ops.OP_N, epilogue=ops.EPI_ADD, aux=dz, precision=P)
    return z, dx, dw1, dc1, dw2, dc2


@pytest.mark.parametrize("M", [1, 31, 32, 33, 64, 65, 1000, 16384 + 7, 70000])
def test_exact_on_integers(M):
    g = torch.Generator().manual_seed(M)
    ri = lambda *s, lo=-2, hi=3: torch.randint(lo, hi, s, generator=g).float()
    x, dz = ri(M, E), ri(M, E)
    w1, c1, w2, c2 = ri(HID, E), ri(HID, lo=-3, hi=4), ri(E, HID), ri(E, lo=-3, hi=4)
    want = _ref(x, w1, c1, w2, c2, dz)
    got = _fused(*(t.cuda() for t in (x, w1, c1, w2, c2, dz)))
    for name, a, b in zip(("z", "dx", "dw1", "dc1", "dw2", "dc2"), got, want):
        if name in ("dw1", "dw2", "dc1", "dc2") and M > 20000:      # sums of ~70 000 integer terms leave the 2^24 exact range of fp32
            torch.testing.assert_close(a.cpu().double(), b, rtol=1e-6, atol=1e-3, msg=name)
        else:
            assert torch.equal(a.cpu().double(), b), name


@pytest.mark.parametrize("M", [4000, 225280])
@pytest.mark.parametrize("kind", ["normal", "cancel", "wide"])
def test_fp32_grade_gate(M, kind):
    g = torch.Generator().manual_seed(M + len(kind))
    x = torch.randn(M, E, generator=g)
    w1, w2 = torch.randn(HID, E, generator=g) * 0.2, torch.randn(E, HID, generator=g) * 0.1
    c1, c2 = torch.randn(HID, generator=g) * 0.1, torch.randn(E, generator=g) * 0.1
    dz = torch.randn(M, E, generator=g)
    if kind == "cancel":          # inner products of large terms that cancel
        x = x * 50
        x[:, 1::2] = -x[:, 0::2] + torch.randn(M, E // 2, generator=g) * 0.02
        w1[:, 1::2] = w1[:, 0::2]
    elif kind == "wide":          # exponents spread over 2^+-12 per element
        x = x * torch.exp2(torch.randint(-12, 13, (M, E), generator=g).float())
        w1 = w1 * torch.exp2(torch.randint(-12, 13, (HID, E), generator=g).float())
    # ReLU' is a step: a pre-activation within rounding of zero flips its mask in ANY fp32 evaluation (the native path's too) and the
    # flipped element is off by the whole gradient, not by a rounding -- which says nothing about the arithmetic.  Tokens with a
    # pre-activation that close to zero are redrawn (about 4 % of them at |pre| < 1e-3 x its scale).
    for _ in range(50):
        pre = x.double().cuda() @ w1.double().cuda().T + c1.double().cuda()
        bad = ((pre.abs() < 1e-3 * pre.abs().mean()).any(dim=1)).cpu()
        if not bool(bad.any()):
            break
        x[bad] = x[torch.randint(0, M, (int(bad.sum()),), generator=g)] * (1 + 0.37 * torch.rand(int(bad.sum()), 1, generator=g))
    assert not bool(bad.any())
    want = _ref(x, w1, c1, w2, c2, dz)
    dev = tuple(t.cuda() for t in (x, w1, c1, w2, c2, dz))
    got, nat = _fused(*dev), _native(*dev)
    for name, w, a, b in zip(("z", "dx", "dw1", "dc1", "dw2", "dc2"), want, nat, got):
        w = w.cuda()
        e_nat, e_fus = (a.double() - w).abs(), (b.double() - w).abs()
        floor = 1e-7 * float(w.abs().max())
        rmax = float(e_fus.max()) / max(float(e_nat.max()), floor)
        rrms = float(e_fus.pow(2).mean().sqrt()) / max(float(e_nat.pow(2).mean().sqrt()), floor / 8)
        assert rmax <= 1.5 and rrms <= 1.5, (name, rmax, rrms)


def test_deterministic_and_strided_rows():
    """Same bits twice; x and dz as column slices of wider matrices (row stride != emb)."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(3)
    M = 50001
    wide_x, wide_d = torch.randn(M, 3 * E, generator=g).cuda(), torch.randn(M, 2 * E, generator=g).cuda()
    x, dz = wide_x[:, E:2 * E], wide_d[:, E:]
    w1, w2 = (torch.randn(HID, E, generator=g) * 0.2).cuda(), (torch.randn(E, HID, generator=g) * 0.1).cuda()
    c1, c2 = torch.randn(HID, generator=g).cuda() * 0.1, torch.randn(E, generator=g).cuda() * 0.1
    a = _fused(x, w1, c1, w2, c2, dz)
    b = _fused(x.contiguous(), w1, c1, w2, c2, dz.contiguous())
    c = _fused(x, w1, c1, w2, c2, dz)
    for u, v, w in zip(a, b, c):
        assert torch.equal(u, v) and torch.equal(u, w)


def test_refuses_other_widths():
    from multimodal_supernovae_amd import ops, _lib
    assert ops.ffn_supported(100, 32, 128) and not ops.ffn_supported(100, 64, 256) and not ops.ffn_supported(100, 32, 192)
    x = torch.zeros(8, 64).cuda()
    w1p, w2tp = ops.ffn_weight_planes(torch.zeros(256, 64).cuda(), torch.zeros(64, 256).cuda())
    with pytest.raises(_lib.MsnHipError):
        ops.ffn_fwd(x, w1p, w2tp, torch.zeros(256).cuda(), torch.zeros(64).cuda())

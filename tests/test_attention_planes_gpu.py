"""fp32-grade attention on the bf16 matrix cores for long sequences of narrow heads (csrc/attention_planes.hip; replaces
SelfAttention of ref src/transformer_utils.py:36-89 at the spectrum / light-curve tower shapes).

THE GATE (the rule the plane GEMMs passed): on every output -- out, dq, dk, dv -- maximum AND RMS error
against the reference's formula in fp64 at most 1.5 x those of the exact-fp32 matrix-core kernels it replaces
(v_mfma_f32_16x16x4_f32, msn_set_attention_planes(0)) on the same inputs: N(0,1) data, masked keys, a fully padded sample,
large-magnitude scores (peaked softmax), cross attention, the ragged ends of every tile size.  The measured ratios go to
gpurun_out/r06_attention_planes_accuracy.txt."""
import math
import os

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


def _report(line):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "r06_attention_planes_accuracy.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def _run(q, k, v, dout, mu8, heads, scale, planes_mode, path):
    from multimodal_supernovae_amd import _lib, ops
    ops.set_attention_planes(planes_mode)
    _lib.check(_lib.lib().msn_set_attention_path(path))
    try:
        out, lse = ops.attention_fwd(q, k, v, mu8, heads, scale)
        dq, dk, dv = (torch.full_like(t, float("nan")) for t in (q, k, v))
        ops.attention_bwd(q, k, v, mu8, heads, scale, out, lse, dout, dq, dk, dv)
    finally:
        ops.set_attention_planes(1)
        _lib.lib().msn_set_attention_path(0)
    return out, lse, dq, dk, dv


CASES = [  # B, Tq, Tk, heads, hd, masked, gain (multiplies q: peaked softmax), path
    (4, 220, 220, 2, 16, False, 1.0, 0),        # Maven spectrum tower (configs/maven-lite.yaml)
    (3, 220, 220, 2, 16, True, 1.0, 0),
    (2, 1024, 1024, 2, 16, False, 1.0, 0),      # 1024-bin spectra
    (2, 1024, 1024, 2, 16, True, 4.0, 0),
    (2, 300, 300, 3, 16, True, 8.0, 0),
    (2, 129, 129, 1, 16, False, 1.0, 0),
    (3, 257, 513, 2, 16, True, 1.0, 0),         # cross attention, ragged chunk ends
    (2, 150, 700, 2, 16, False, 2.0, 0),
    (2, 200, 200, 8, 8, True, 1.0, 2),          # the light-curve tower's 8-wide heads, run as 16-wide ones (path 2)
    (2, 333, 333, 2, 12, True, 1.0, 2),
    (1, 640, 640, 4, 4, False, 1.0, 2),
]


# 3 = the two-kernel backward (dQ, then dK,dV: what few (sample, head) pairs take by default); 5 = the one-pass backward (what 256 or
# more pairs take by default: every training shape)
@pytest.mark.parametrize("mode", [3, 5])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(x) for x in c))
def test_fp32_grade_gate(case, mode):
    from multimodal_supernovae_amd import ops
    B, Tq, Tk, heads, hd, masked, gain, path = case
    E = heads * hd
    g = torch.Generator().manual_seed(B * 1000 + Tq + Tk + hd)
    q = torch.randn(B, Tq, E, generator=g) * gain
    k, v, dout = torch.randn(B, Tk, E, generator=g), torch.randn(B, Tk, E, generator=g), torch.randn(B, Tq, E, generator=g)
    mask = None
    if masked:
        mask = torch.rand(B, Tk, generator=g) > 0.3
        mask[:, 0] = True
        mask[0, -1] = False
        if B > 1:
            mask[-1] = False                    # a fully padded sample: uniform attention, zero score gradients
    scale = 1.0 / math.sqrt(E)
    qr, kr, vr = (t.double().cuda().requires_grad_() for t in (q, k, v))
    ref = _ref(qr, kr, vr, mask.cuda() if masked else None, heads, scale)
    ref.backward(dout.double().cuda())
    want = (ref.detach(), qr.grad, kr.grad, vr.grad)
    qc, kc, vc, dc = q.cuda(), k.cuda(), v.cuda(), dout.cuda()
    mu8 = ops._mask_u8(mask.cuda()) if masked else None
    nat = _run(qc, kc, vc, dc, mu8, heads, scale, 0, path)
    pl = _run(qc, kc, vc, dc, mu8, heads, scale, mode, path)
    assert not any(torch.isnan(t).any() for t in pl), "an output element was not written"
    # the row statistics the backward kernels read: same layout, same values to rounding
    torch.testing.assert_close(pl[1][..., 0], nat[1][..., 0], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(pl[1][..., 1], nat[1][..., 1], rtol=1e-5, atol=1e-5)
    for name, w, a, b in zip(("out", "dq", "dk", "dv"), want, (nat[0], *nat[2:]), (pl[0], *pl[2:])):
        e_nat, e_pl = (a.double() - w).abs(), (b.double() - w).abs()
        floor = 1e-7 * float(w.abs().max())               # where both are exact to rounding of the result itself
        rmax = float(e_pl.max()) / max(float(e_nat.max()), floor)
        rrms = float(e_pl.pow(2).mean().sqrt()) / max(float(e_nat.pow(2).mean().sqrt()), floor / 8)
        _report(f"mode {mode} case {case} {name}: planes / fp32-MFMA error  max {rmax:.3f}  rms {rrms:.3f}   "
                f"(abs max planes {float(e_pl.max()):.3e}, fp32 {float(e_nat.max()):.3e}, |ref| max {float(w.abs().max()):.3e})")
        assert rmax <= 1.5 and rrms <= 1.5, (name, rmax, rrms)


def test_statistics_feed_either_backward():
    """The row statistics the plane forward writes are the exact-fp32 kernels' statistics: either backward may follow either
    forward (a checkpoint / a graph recorded under the other setting)."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, heads, hd = 2, 220, 2, 16
    q, k, v, dout = (torch.randn(B, T, heads * hd, generator=g).cuda() for _ in range(4))
    scale = 1.0 / math.sqrt(heads * hd)
    ops.set_attention_planes(1)
    out, lse = ops.attention_fwd(q, k, v, None, heads, scale)
    ref = _run(q, k, v, dout, None, heads, scale, 0, 0)
    ops.set_attention_planes(0)
    try:
        dq, dk, dv = (torch.empty_like(t) for t in (q, k, v))
        ops.attention_bwd(q, k, v, None, heads, scale, out, lse, dout, dq, dk, dv)
    finally:
        ops.set_attention_planes(1)
    for a, b in zip((dq, dk, dv), ref[2:]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)


def test_deterministic_and_strided():
    """Same bits twice; q | k | v as column slices of one packed (B, T, 3E) buffer (the towers' layout), gradients written into
    slices of a packed buffer."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(6)
    B, T, heads, hd = 3, 1024, 2, 16
    E = heads * hd
    qkv = torch.randn(B, T, 3 * E, generator=g).cuda()
    dout = torch.randn(B, T, E, generator=g).cuda()
    mask = torch.rand(B, T, generator=g) > 0.2
    mask[:, 0] = True
    mu8 = ops._mask_u8(mask.cuda())
    scale = 1.0 / math.sqrt(E)
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    res = []
    for _ in range(2):
        out, lse = ops.attention_fwd(q, k, v, mu8, heads, scale)
        d = torch.full_like(qkv, float("nan"))
        ops.attention_bwd(q, k, v, mu8, heads, scale, out, lse, dout, d[..., :E], d[..., E:2 * E], d[..., 2 * E:])
        res.append((out, lse, d))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert not torch.isnan(res[0][2]).any()
    qc, kc, vc = (t.contiguous() for t in (q, k, v))
    out2, _, dq, dk, dv = _run(qc, kc, vc, dout, mu8, heads, scale, 1, 0)
    assert torch.equal(out2, res[0][0]) and torch.equal(torch.cat([dq, dk, dv], -1), res[0][2])


@pytest.mark.parametrize("B,T,heads,hd,masked", [(160, 300, 2, 16, True), (130, 1024, 2, 16, False), (64, 220, 4, 12, True)])
def test_one_pass_backward_at_training_batches(B, T, heads, hd, masked):
    """256 or more (sample, head) pairs take pattn_bwd_fused_kernel by default: one workgroup per pair walks every key block and sums
    dq over them in memory.  Same bits twice (the accumulation order is fixed), no element left unwritten, dq / dk / dv equal to the
    two-kernel backward's to rounding, gradients into column slices of a packed buffer."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(B + T)
    E = heads * hd
    qkv = torch.randn(B, T, 3 * E, generator=g).cuda()
    dout = torch.randn(B, T, E, generator=g).cuda()
    mu8 = None
    if masked:
        mask = torch.rand(B, T, generator=g) > 0.25
        mask[:, 0] = True
        mask[-1] = False
        mu8 = ops._mask_u8(mask.cuda())
    scale = 1.0 / math.sqrt(E)
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    out, lse = ops.attention_fwd(q, k, v, mu8, heads, scale)
    res = []
    for mode in (1, 1, 3):
        ops.set_attention_planes(mode)
        try:
            d = torch.full_like(qkv, float("nan"))
            ops.attention_bwd(q, k, v, mu8, heads, scale, out, lse, dout, d[..., :E], d[..., E:2 * E], d[..., 2 * E:])
        finally:
            ops.set_attention_planes(1)
        res.append(d)
    assert not torch.isnan(res[0]).any()
    assert torch.equal(res[0], res[1])
    torch.testing.assert_close(res[0], res[2], rtol=2e-5, atol=2e-5 * float(res[2].abs().max()))



@pytest.mark.parametrize("B,T,heads,hd,masked", [(40, 200, 8, 8, True), (64, 333, 4, 12, False)])
def test_narrow_heads_vector_forward_plane_backward(B, T, heads, hd, masked):
    """Heads narrower than 16 over more than 128 tokens (the light-curve tower: 8 x 8, 200 steps) keep the vector-ALU forward and,
    from 256 (sample, head) pairs on, take the one-pass plane backward behind it (msn_attention_bwd's default route): the row
    statistics of the two families are the same (max, log-sum) pairs.  Gate as above, against the vector-ALU backward it replaces."""
    from multimodal_supernovae_amd import ops
    E = heads * hd
    g = torch.Generator().manual_seed(B + T + hd)
    q, k, v, dout = (torch.randn(B, T, E, generator=g) for _ in range(4))
    mask = None
    if masked:
        mask = torch.rand(B, T, generator=g) > 0.3
        mask[:, 0] = True
        mask[-1] = False
    scale = 1.0 / math.sqrt(E)
    qr, kr, vr = (t.double().cuda().requires_grad_() for t in (q, k, v))
    ref = _ref(qr, kr, vr, mask.cuda() if masked else None, heads, scale)
    ref.backward(dout.double().cuda())
    want = (qr.grad, kr.grad, vr.grad)
    qc, kc, vc, dc = q.cuda(), k.cuda(), v.cuda(), dout.cuda()
    mu8 = ops._mask_u8(mask.cuda()) if masked else None
    valu = _run(qc, kc, vc, dc, mu8, heads, scale, 0, 0)          # planes off: vector-ALU forward and backward
    mixed = _run(qc, kc, vc, dc, mu8, heads, scale, 1, 0)         # default: vector-ALU forward, plane backward
    assert torch.equal(valu[0], mixed[0]) and torch.equal(valu[1], mixed[1])      # the same forward
    assert not any(torch.isnan(t).any() for t in mixed)
    for name, w, a, b in zip(("dq", "dk", "dv"), want, valu[2:], mixed[2:]):
        e_v, e_m = (a.double() - w).abs(), (b.double() - w).abs()
        floor = 1e-7 * float(w.abs().max())
        rmax = float(e_m.max()) / max(float(e_v.max()), floor)
        rrms = float(e_m.pow(2).mean().sqrt()) / max(float(e_v.pow(2).mean().sqrt()), floor / 8)
        _report(f"narrow heads {(B, T, heads, hd, masked)} {name}: plane backward / vector-ALU backward error  max {rmax:.3f}  rms {rrms:.3f}")
        assert rmax <= 1.5 and rrms <= 1.5, (name, rmax, rrms)

"""GPU parity of the fused InfoNCE kernels (through the C-ABI) against the oracle, the golden
vectors from the reference, and size-independent properties at the BASELINE sizes.
Tolerance: 1e-3 relative (north star); in practice the fp32 kernels sit near 1e-5."""
import math

import pytest
import torch

from conftest import Fixture, golden_names

pytestmark = pytest.mark.gpu
RTOL = 1e-3


def _unit(n, d, seed):
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(seed))
    return x / x.norm(dim=-1, keepdim=True)


def _run_hip(e1, e2, ls, lb):
    from multimodal_supernovae_amd.loss import clip_loss
    a, b = e1.cuda().requires_grad_(), e2.cuda().requires_grad_()
    s, c = ls.cuda().requires_grad_(), lb.cuda().requires_grad_()
    loss = clip_loss(a, b, s, c)
    loss.backward()
    return loss.detach().cpu(), a.grad.cpu(), b.grad.cpu(), s.grad.cpu(), c.grad.cpu()


def _run_oracle(e1, e2, ls, lb):
    from oracle.loss import clip_loss
    a, b = e1.double().requires_grad_(), e2.double().requires_grad_()
    s, c = ls.double().requires_grad_(), lb.double().requires_grad_()
    loss = clip_loss(a, b, s, c)
    loss.backward()
    return loss.detach(), a.grad, b.grad, s.grad, c.grad


def _compare(hip, ref, n):
    loss, d1, d2, ds, db = hip
    rl, r1, r2, rs, rb = ref
    assert abs(float(loss) - float(rl)) <= RTOL * abs(float(rl)) + 1e-6
    gscale = float(r1.abs().max())
    torch.testing.assert_close(d1.double(), r1, rtol=RTOL, atol=RTOL * gscale)
    torch.testing.assert_close(d2.double(), r2, rtol=RTOL, atol=RTOL * gscale)
    assert abs(float(ds) - float(rs)) <= RTOL * abs(float(rs)) + 1e-5
    assert abs(float(db)) <= 1e-4, "logit_bias is gradient-free under the softmax loss"


@pytest.mark.parametrize("name", golden_names("loss_clip_n"))
def test_against_reference_golden(name):
    f = Fixture(name)
    i = f.groups["in"]
    hip = _run_hip(i["e1"], i["e2"], i["logit_scale"], i["logit_bias"])
    ref = (f.out["loss"].double(), f.grad["e1"].double(), f.grad["e2"].double(), f.grad["logit_scale"].double(),
           f.grad["logit_bias"].double())
    _compare(hip, ref, i["e1"].shape[0])


@pytest.mark.parametrize("n,d", [(1, 8), (5, 16), (32, 128), (33, 32), (129, 64), (256, 128), (1024, 128),
                                 (1000, 128), (4096, 128),
                                 # any enc_dim the reference accepts (src/models_multimodal.py:101): zero-padded tiles
                                 (40, 1), (77, 5), (64, 12), (300, 100), (130, 130), (257, 200), (96, 256)])
@pytest.mark.parametrize("log_scale,bias", [(math.log(10.0), -10.0), (math.log(31.0), 0.5)])
def test_against_oracle(n, d, log_scale, bias):
    e1, e2 = _unit(n, d, 100 + n), _unit(n, d, 200 + n)
    ls, lb = torch.tensor(log_scale), torch.tensor(bias)
    _compare(_run_hip(e1, e2, ls, lb), _run_oracle(e1, e2, ls, lb), n)


def test_known_answers():
    from multimodal_supernovae_amd.loss import clip_loss
    f = Fixture("loss_kat")
    eye = torch.eye(4)
    pad = torch.zeros(4, 4)
    e = torch.cat([eye, pad], dim=1).cuda()          # D = 8 (kernel granule), same dot products as I4
    a = clip_loss(e, e, torch.tensor(0.0), torch.tensor(0.0))
    assert abs(float(a) - float(f.out["clip_I4_s0_b0"])) < 1e-5
    b = clip_loss(e, e, torch.tensor(math.log(10.0)), torch.tensor(-10.0))
    assert abs(float(b) - float(f.out["clip_I4_ln10_bm10"])) < 1e-6


def test_unequal_lengths_match_reference():
    from multimodal_supernovae_amd.loss import clip_loss
    f = Fixture("loss_clip_unequal")
    i = f.groups["in"]
    loss = clip_loss(i["e1"].cuda(), i["e2"].cuda(), i["logit_scale"].cuda(), i["logit_bias"].cuda())
    assert abs(float(loss) - float(f.out["loss"])) <= RTOL * abs(float(f.out["loss"]))
    # gradients of the ragged case against the oracle
    e1, e2 = _unit(70, 32, 1), _unit(45, 32, 2)
    ls, lb = torch.tensor(1.1), torch.tensor(-0.3)
    hip, ref = _run_hip(e1, e2, ls, lb), _run_oracle(e1, e2, ls, lb)
    torch.testing.assert_close(hip[1].double(), ref[1], rtol=RTOL, atol=1e-6)
    torch.testing.assert_close(hip[2].double(), ref[2], rtol=RTOL, atol=1e-6)
    assert abs(float(hip[3]) - float(ref[3])) <= RTOL * abs(float(ref[3])) + 1e-6


def test_three_way_multimodal_matches_reference():
    from multimodal_supernovae_amd.loss import clip_loss_multimodal
    f = Fixture("loss_clip_multimodal3")
    i = f.groups["in"]
    embs = [i[k].cuda().requires_grad_() for k in ("e0", "e1", "e2")]
    ls, lb = i["logit_scale"].cuda().requires_grad_(), i["logit_bias"].cuda().requires_grad_()
    loss = clip_loss_multimodal(embs, ls, lb)
    loss.backward()
    assert abs(float(loss) - float(f.out["loss"])) <= RTOL * abs(float(f.out["loss"]))
    for k, e in zip(("e0", "e1", "e2"), embs):
        torch.testing.assert_close(e.grad.cpu(), f.grad[k], rtol=RTOL, atol=1e-6)
    assert abs(float(ls.grad) - float(f.grad["logit_scale"])) <= RTOL * abs(float(f.grad["logit_scale"])) + 1e-6
    lv = clip_loss_multimodal([e.detach() for e in embs], i["scales_vec"].cuda(), i["biases_vec"].cuda())
    assert abs(float(lv) - float(f.out["loss_vec"])) <= RTOL * abs(float(f.out["loss_vec"]))


def test_strided_inputs():
    """Embeddings handed over as column slices of a wider (B, M*D) buffer (the all-gather layout)."""
    from multimodal_supernovae_amd.loss import clip_loss
    both = torch.cat([_unit(200, 64, 7), _unit(200, 64, 8)], dim=1).cuda()
    e1, e2 = both[:, :64], both[:, 64:]
    ls, lb = torch.tensor(2.0).cuda(), torch.tensor(-1.0).cuda()
    a = clip_loss(e1, e2, ls, lb)
    b = clip_loss(e1.contiguous(), e2.contiguous(), ls, lb)
    assert float(a) == float(b)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_row_sharded_kernels_sum_to_single_process(world):
    """Emulate `world` ranks on one GPU: per-shard kernel calls with q_offset, LSEs concatenated
    (the all-gather), shares summed (the all-reduce) == the unsharded call; also == the dense oracle."""
    from multimodal_supernovae_amd.loss import HipPairKernels as K
    from oracle.sharded import OraclePairKernels as O
    b, d = 96, 128
    n = b * world
    e1, e2 = _unit(n, d, 31).cuda(), _unit(n, d, 32).cuda()
    ls, lb = torch.tensor(math.log(19.5)).cuda(), torch.tensor(-10.0).cuda()
    one = torch.tensor(1.0).cuda()
    lr, lc, total = K.forward(e1, e2, e1, e2, 0, ls, lb)
    g1, g2, gs, gb = K.backward(e1, e2, e1, e2, 0, ls, lb, lr, lc, one)
    parts = [K.forward(e1[r * b:(r + 1) * b], e2[r * b:(r + 1) * b], e1, e2, r * b, ls, lb) for r in range(world)]
    lr_all = torch.cat([p[0] for p in parts])
    lc_all = torch.cat([p[1] for p in parts])
    torch.testing.assert_close(lr_all, lr, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lc_all, lc, rtol=1e-5, atol=1e-5)
    assert abs(float(sum(p[2] for p in parts)) - float(total)) < 1e-5
    ds_sum = 0.0
    for r in range(world):
        sl = slice(r * b, (r + 1) * b)
        d1, d2, ds, db = K.backward(e1[sl], e2[sl], e1, e2, r * b, ls, lb, lr_all, lc_all, one)
        torch.testing.assert_close(d1, g1[sl], rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(d2, g2[sl], rtol=1e-4, atol=1e-7)
        o1, o2, os_, ob = O.backward(e1[sl], e2[sl], e1, e2, r * b, ls, lb, lr_all, lc_all, one)
        torch.testing.assert_close(d1, o1, rtol=RTOL, atol=1e-6)
        torch.testing.assert_close(d2, o2, rtol=RTOL, atol=1e-6)
        ds_sum += float(ds)
    assert abs(ds_sum - float(gs)) <= 1e-4 * abs(float(gs)) + 1e-6


def test_properties_at_full_size():
    """N = 4096 (BASELINE cfg5 global batch): symmetry in the two modalities, invariance to the
    bias (it cancels in both log-softmaxes), determinism, loss(identical unit vectors) bound."""
    from multimodal_supernovae_amd.loss import clip_loss
    n, d = 4096, 128
    e1, e2 = _unit(n, d, 5).cuda(), _unit(n, d, 6).cuda()
    ls = torch.tensor(math.log(19.545966923442453)).cuda()
    a = clip_loss(e1, e2, ls, torch.tensor(-10.0).cuda())
    b = clip_loss(e2, e1, ls, torch.tensor(-10.0).cuda())
    c = clip_loss(e1, e2, ls, torch.tensor(3.0).cuda())
    assert abs(float(a) - float(b)) < 1e-5 * abs(float(a))
    assert abs(float(a) - float(c)) < 1e-4 * abs(float(a))
    assert float(clip_loss(e1, e2, ls, torch.tensor(-10.0).cuda())) == float(a)
    same = clip_loss(e1, e1, ls, torch.tensor(0.0).cuda())
    assert float(same) < float(a)
    assert float(same) <= math.log(n)


def test_rejects_unsupported_width_and_cpu_tensors():
    from multimodal_supernovae_amd import _lib
    from multimodal_supernovae_amd.loss import clip_loss
    with pytest.raises(_lib.MsnHipError):        # D > 256 is the one width the tiles do not take
        clip_loss(_unit(8, 260, 1).cuda(), _unit(8, 260, 2).cuda(), torch.tensor(0.0).cuda(), torch.tensor(0.0).cuda())
    with pytest.raises(_lib.MsnHipError):        # embeddings on the host: the product has no CPU path
        clip_loss(_unit(8, 16, 1), _unit(8, 16, 2), torch.tensor(0.0), torch.tensor(0.0))


def test_odd_width_strided_and_unaligned():
    """D = 100 as column slices of the packed all-gather buffer (row stride 300, second / third blocks start at
    columns 100 / 200: 16-byte aligned, third pair shifted by one float: not aligned)."""
    from multimodal_supernovae_amd.loss import clip_loss
    wide = torch.cat([_unit(150, 100, 1), _unit(150, 100, 2), _unit(150, 101, 3)], dim=1).cuda()
    ls, lb = torch.tensor(2.5).cuda(), torch.tensor(-1.0).cuda()
    from multimodal_supernovae_amd.loss import HipPairKernels as K
    one = torch.tensor(1.0).cuda()
    for c1, c2 in ((0, 100), (100, 201)):
        e1, e2 = wide[:, c1:c1 + 100], wide[:, c2:c2 + 100]
        c1_, c2_ = e1.contiguous(), e2.contiguous()
        lr, lc, a = K.forward(e1, e2, e1, e2, 0, ls, lb)          # row-strided views straight into the C-ABI
        lr2, lc2, b = K.forward(c1_, c2_, c1_, c2_, 0, ls, lb)
        assert float(a) == float(b) and torch.equal(lr, lr2) and torch.equal(lc, lc2)
        assert float(a) == float(clip_loss(e1, e2, ls, lb))
        ga = K.backward(e1, e2, e1, e2, 0, ls, lb, lr, lc, one)
        gb = K.backward(c1_, c2_, c1_, c2_, 0, ls, lb, lr, lc, one)
        for x, y in zip(ga, gb):
            assert torch.equal(x, y)


# ------------------------------------------------------------------------------------ sigmoid loss
@pytest.mark.parametrize("name", golden_names("loss_sigmoid_n"))
def test_sigmoid_loss_against_reference_golden(name):
    from multimodal_supernovae_amd.loss import sigmoid_loss
    f = Fixture(name)
    i = f.groups["in"]
    a, b = i["e1"].cuda().requires_grad_(), i["e2"].cuda().requires_grad_()
    s, c = i["logit_scale"].cuda().requires_grad_(), i["logit_bias"].cuda().requires_grad_()
    loss = sigmoid_loss(a, b, s, c)
    loss.backward()
    assert abs(float(loss.detach()) - float(f.out["loss"])) <= RTOL * abs(float(f.out["loss"]))
    for got, k in ((a.grad, "e1"), (b.grad, "e2")):
        ref = f.grad[k].float()
        torch.testing.assert_close(got.cpu(), ref, rtol=RTOL, atol=RTOL * float(ref.abs().max()))
    assert abs(float(s.grad) - float(f.grad["logit_scale"])) <= RTOL * abs(float(f.grad["logit_scale"])) + 1e-6
    assert abs(float(c.grad) - float(f.grad["logit_bias"])) <= RTOL * abs(float(f.grad["logit_bias"])) + 1e-6


def test_sigmoid_known_answer_and_multimodal():
    from multimodal_supernovae_amd.loss import sigmoid_loss, sigmoid_loss_multimodal
    kat = Fixture("loss_kat")
    e = torch.cat([torch.eye(4), torch.zeros(4, 4)], dim=1).cuda()
    v = sigmoid_loss(e, e, torch.tensor(math.log(10.0)), torch.tensor(-10.0))
    assert abs(float(v) - float(kat.out["sigmoid_I4_ln10_bm10"])) < 1e-4          # 7.5000340
    f = Fixture("loss_clip_multimodal3")
    i = f.groups["in"]
    embs = [i[k].cuda() for k in ("e0", "e1", "e2")]
    got = sigmoid_loss_multimodal(embs, i["logit_scale"].cuda(), i["logit_bias"].cuda())
    assert abs(float(got) - float(f.out["sigmoid"])) <= RTOL * abs(float(f.out["sigmoid"]))


@pytest.mark.parametrize("n,d", [(33, 32), (256, 128), (1024, 128), (70, 100)])
def test_sigmoid_loss_against_oracle(n, d):
    from multimodal_supernovae_amd.loss import sigmoid_loss
    from oracle.loss import sigmoid_loss as ref_fn
    e1, e2 = _unit(n, d, 300 + n), _unit(n, d, 400 + n)
    ls, lb = torch.tensor(math.log(5.0)), torch.tensor(-3.0)
    a, b = e1.cuda().requires_grad_(), e2.cuda().requires_grad_()
    s, c = ls.cuda().requires_grad_(), lb.cuda().requires_grad_()
    sigmoid_loss(a, b, s, c).backward()
    ra, rb = e1.clone().requires_grad_(), e2.clone().requires_grad_()
    rs, rc = ls.clone().requires_grad_(), lb.clone().requires_grad_()
    ref_fn(ra, rb, rs, rc).backward()
    for got, ref in ((a.grad, ra.grad), (b.grad, rb.grad)):
        torch.testing.assert_close(got.cpu(), ref, rtol=RTOL, atol=RTOL * float(ref.abs().max()))
    assert abs(float(s.grad) - float(rs.grad)) <= RTOL * abs(float(rs.grad)) + 1e-6
    assert abs(float(c.grad) - float(rc.grad)) <= RTOL * abs(float(rc.grad)) + 1e-6


# -------------------------------------------------------------------------------- retrieval AUC (row f2)
def test_retrieval_auc_matches_reference_golden():
    import numpy as np
    from multimodal_supernovae_amd.utils import get_AUC, get_ROC_data
    f = Fixture("auc")
    for n in (50, 137):
        e1, e2 = f.groups["in"][f"e1_{n}"].cuda(), f.groups["in"][f"e2_{n}"].cuda()
        t, frac = get_ROC_data(e1, e2)
        assert np.allclose(t, f.out[f"thresholds_{n}"].numpy()) and np.allclose(frac, f.out[f"fraction_{n}"].numpy())
        assert abs(get_AUC(e1, e2) - float(f.out[f"auc_{n}"])) < 1e-12


@pytest.mark.parametrize("n,d", [(1000, 128), (4096, 128), (33, 8), (500, 100)])
def test_retrieval_ranks_against_oracle(n, d):
    from multimodal_supernovae_amd.utils import retrieval_ranks
    from oracle.clip import roc_data
    g = torch.Generator().manual_seed(n)
    e1 = torch.randn(n, d, generator=g)
    e2 = e1 + 2.0 * torch.randn(n, d, generator=g)
    got = retrieval_ranks(e1.cuda(), e2.cuda()).cpu().numpy()
    _, _, ref = roc_data(e1.double(), e2.double())
    assert (abs(got - ref) <= 1).all() and (got != ref).mean() < 0.01      # fp32 vs fp64 near-ties only

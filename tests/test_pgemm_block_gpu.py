"""The plane path of the build-defined ViT block (functional._PreNormBlockPlanes: every Linear product on the bf16 matrix
cores from resident bf16 planes) against the fp32 block: output and every gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision,tol", [("bf16x6", 2e-5), ("bf16x3p", 2e-3)])
def test_block_matches_fp32_block(precision, tol):
    from multimodal_supernovae_amd import functional as F_, ops
    torch.manual_seed(0)
    B, T, e, heads = 8, 65, 384, 6
    g = torch.Generator().manual_seed(1)
    def rnd(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).cuda().requires_grad_()
    p = [rnd(e).detach().add_(1).requires_grad_(), rnd(e, scale=0.1), rnd(3 * e, e, scale=0.05), rnd(3 * e, scale=0.1),
         rnd(e, e, scale=0.05), rnd(e, scale=0.1), rnd(e).detach().add_(1).requires_grad_(), rnd(e, scale=0.1),
         rnd(4 * e, e, scale=0.05), rnd(4 * e, scale=0.1), rnd(e, 4 * e, scale=0.05), rnd(e, scale=0.1)]
    x = rnd(B, T, e)
    dy = torch.randn(B, T, e, generator=g).cuda()
    def run(prec):
        for t in p + [x]:
            t.grad = None
        with ops.gemm_precision(prec):
            y = F_.pre_norm_block(x, heads, p, eps=1e-6)
            y.backward(dy)
        return [y.detach().clone()] + [t.grad.clone() for t in [x] + p]
    ref = run("f32")
    got = run(precision)
    for i, (a, b) in enumerate(zip(got, ref)):
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        assert err < tol, (i, err)

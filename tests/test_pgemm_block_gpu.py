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


BIG = 32 * 1024      # rows from which the row-block kernels (ln_fwd_planes_kernel / ln_bwd_planes_kernel) take 260..400 columns


@pytest.mark.parametrize("rows,cols,planes", [(260, 384, 3), (77, 64, 3), (1000, 1024, 2), (33, 200, 3), (4161, 384, 3), (95, 320, 2),
                                              (64, 400, 3), (1, 264, 3), (1055, 384, 2), (40000, 384, 3),
                                              (BIG + 260, 384, 3), (BIG + 33, 320, 2), (BIG + 1, 264, 3), (BIG, 400, 3), (BIG + 95, 384, 2)])
def test_layernorm_plane_outputs(rows, cols, planes):
    """LayerNorm forward / backward writing planes == the fp32 kernels' results split by msn_plane_split, bit for bit: the
    row-at-a-time kernels (below 32 768 rows, and every width outside 260..400 columns) and the row-block kernels (from 32 768 rows
    on: ragged last row block, a padding column block, both plane counts)."""
    from multimodal_supernovae_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g).cuda()
    gm, bt = (torch.randn(cols, generator=g) + 1).cuda(), torch.randn(cols, generator=g).cuda()
    dy, add = torch.randn(rows, cols, generator=g).cuda(), torch.randn(rows, cols, generator=g).cuda()
    y, mean, rstd = ops.layernorm_fwd(x, gm, bt, 1e-6)
    yp, mean2, rstd2 = ops.layernorm_fwd_planes(x, gm, bt, 1e-6, planes)
    assert torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    assert torch.equal(yp.buf, ops.plane_split(y, planes).buf)
    dx, dg, db = ops.layernorm_bwd(dy, x, mean, rstd, gm, add=add)
    dx2, dxp, dg2, db2, cs = ops.layernorm_bwd_planes(dy, x, mean, rstd, gm, planes, add=add, want_colsum=True)
    assert torch.equal(dx, dx2)
    assert torch.equal(dxp.buf, ops.plane_split(dx, planes).buf)
    block_kernel = rows >= BIG and 260 <= cols <= 400
    if not block_kernel:
        assert torch.equal(dg, dg2) and torch.equal(db, db2)
    else:                      # the same terms summed in another (fixed) order: against fp64
        xh = (x.double() - mean.double()[:, None]) * rstd.double()[:, None]
        torch.testing.assert_close(dg2.double(), (dy.double() * xh).sum(0), rtol=1e-5, atol=2e-3)
        torch.testing.assert_close(db2.double(), dy.double().sum(0), rtol=1e-5, atol=2e-3)
    torch.testing.assert_close(cs.cpu().double(), dx.cpu().double().sum(0), rtol=1e-5, atol=2e-3)
    if block_kernel:           # twice the same call: the same bytes (fixed-order partial sums)
        again = ops.layernorm_bwd_planes(dy, x, mean, rstd, gm, planes, add=add, want_colsum=True)
        assert torch.equal(again[2], dg2) and torch.equal(again[3], db2) and torch.equal(again[4], cs)


def test_trunk_of_three_blocks():
    """functional.plane_vit_trunk over several blocks (planes handed from block to block) == the fp32 blocks."""
    from multimodal_supernovae_amd import functional as F_, ops
    B, T, e, heads, nb = 8, 65, 384, 6, 3
    g = torch.Generator().manual_seed(5)
    def rnd(*s, scale=1.0, shift=0.0):
        return (torch.randn(*s, generator=g) * scale + shift).cuda().requires_grad_()
    blocks = [[rnd(e, shift=1.0), rnd(e, scale=0.1), rnd(3 * e, e, scale=0.05), rnd(3 * e, scale=0.1), rnd(e, e, scale=0.05),
               rnd(e, scale=0.1), rnd(e, shift=1.0), rnd(e, scale=0.1), rnd(4 * e, e, scale=0.05), rnd(4 * e, scale=0.1),
               rnd(e, 4 * e, scale=0.05), rnd(e, scale=0.1)] for _ in range(nb)]
    x = rnd(B, T, e)
    dy = torch.randn(B, T, e, generator=g).cuda()
    leaves = [x] + [p for blk in blocks for p in blk]
    def run(prec):
        for t in leaves:
            t.grad = None
        with ops.gemm_precision(prec):
            if prec == "f32":
                h = x
                for blk in blocks:
                    h = F_.pre_norm_block(h, heads, blk, eps=1e-6)
            else:
                h = F_.plane_vit_trunk(x, heads, 1e-6, blocks)
            h.backward(dy)
        return [h.detach().clone()] + [t.grad.clone() for t in leaves]
    ref, got = run("f32"), run("bf16x6")
    for i, (a, b) in enumerate(zip(got, ref)):
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        assert err < 5e-5, (i, err)

"""GPU parity of the build-defined ResNet-18 and 1-D CNN encoders and their conv plumbing against the
CPU specification (oracle/build_defined.py) / plain torch.  Not in the reference: parity unpinned by it."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(a, b, what, rtol=1e-3):
    scale = float(b.abs().max()) + 1e-6
    torch.testing.assert_close(a, b.to(a.dtype), rtol=rtol, atol=rtol * scale * 0.1, msg=lambda m: f"{what}: {m}")


@pytest.mark.parametrize("H,W,C,co,k,s,p", [(8, 8, 4, 8, 3, 1, 1), (9, 7, 3, 5, 3, 2, 1), (16, 16, 3, 8, 7, 2, 3),
                                           (6, 6, 8, 16, 1, 2, 0), (5, 5, 4, 4, 1, 1, 0), (1, 20, 4, 8, 5, 1, 2),
                                           (12, 10, 16, 8, 3, 2, 1), (7, 7, 64, 32, 3, 1, 1), (1, 33, 8, 4, 7, 2, 3)])
def test_conv_channels_last_matches_torch(H, W, C, co, k, s, p):
    from multimodal_supernovae_amd import functional as F_
    g = torch.Generator().manual_seed(H * W + C + k)
    kh = 1 if H == 1 else k
    ph = 0 if H == 1 else p
    x = torch.randn(3, C, H, W, generator=g)
    w = torch.randn(co, C, kh, k, generator=g) * 0.3
    b = torch.randn(co, generator=g)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = torch.relu(F.conv2d(xr, wr, br, stride=(1 if H == 1 else s, s), padding=(ph, p)))
    cot = torch.randn(ref.shape, generator=g)
    ref.backward(cot)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
    wg, bg = w.cuda().requires_grad_(), b.cuda().requires_grad_()
    y = F_.conv_cl(xg, wg, bg, (1 if H == 1 else s, s), (ph, p), relu=True)
    close(y.detach().cpu().permute(0, 3, 1, 2), ref.detach(), "y")
    y.backward(cot.permute(0, 2, 3, 1).contiguous().cuda())
    close(xg.grad.cpu().permute(0, 3, 1, 2), xr.grad, "dx")
    close(wg.grad.cpu(), wr.grad, "dw")
    close(bg.grad.cpu(), br.grad, "db")


@pytest.mark.parametrize("B,H,W,C,co,k,s,p,bias", [
    (4, 8, 8, 64, 64, 3, 1, 1, False), (8, 16, 16, 64, 128, 3, 2, 1, False), (4, 8, 8, 128, 64, 3, 1, 1, True),
    (2, 1, 64, 64, 128, 5, 1, 2, True), (32, 4, 4, 256, 256, 3, 1, 1, False), (32, 5, 7, 64, 96, 3, 1, 1, True),
    (32, 6, 10, 96, 160, 3, 2, 1, True), (64, 3, 5, 64, 64, 3, 1, 1, False), (8, 12, 12, 64, 64, 5, 1, 2, False),
    (520, 8, 8, 64, 64, 3, 1, 1, False), (4, 1, 1024, 64, 128, 5, 1, 2, True), (2, 600, 8, 64, 64, 3, 1, 1, False)])
def test_conv_implicit_gemm_matches_torch(B, H, W, C, co, k, s, p, bias):
    """Channel counts that are multiples of 32 run without a column matrix (msn_conv2d_fwd / _dgrad / _wgrad: the GEMM lanes
    gather from the image, padding taps read zeros): y, dx, dw, db against torch in float64, ragged row tiles, odd image
    widths (the pixel index is split with multiply-shift division), strides, 1-D series, the last case with a tail
    of K-slabs and more than one split of the weight gradient."""
    from multimodal_supernovae_amd import functional as F_, ops
    kh, ph, sh = (1, 0, 1) if H == 1 else (k, p, s)
    assert ops.conv2d_implicit_ok(B, H, W, C, co, kh, k, sh, s, ph, p)
    g = torch.Generator().manual_seed(B + H * W + C + k)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(co, C, kh, k, generator=g) * 0.1
    b = torch.randn(co, generator=g) if bias else None
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    br = b.double().requires_grad_() if bias else None
    ref = torch.relu(F.conv2d(xr, wr, br, stride=(sh, s), padding=(ph, p)))
    cot = torch.randn(ref.shape, generator=g)
    ref.backward(cot.double())
    xg = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
    wg = w.cuda().requires_grad_()
    bg = b.cuda().requires_grad_() if bias else None
    calls = []
    real = ops.conv2d_fwd
    try:
        ops.conv2d_fwd = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
        y = F_.conv_cl(xg, wg, bg, (sh, s), (ph, p), relu=True)
    finally:
        ops.conv2d_fwd = real
    assert calls, "the implicit path did not run"
    close(y.detach().cpu().permute(0, 3, 1, 2), ref.detach(), "y")
    y.backward(cot.permute(0, 2, 3, 1).contiguous().cuda())
    close(xg.grad.cpu().permute(0, 3, 1, 2), xr.grad, "dx")
    close(wg.grad.cpu(), wr.grad, "dw")
    if bias:
        close(bg.grad.cpu(), br.grad, "db")


@pytest.mark.parametrize("H,C,co,s", [(16, 64, 64, 1), (16, 64, 128, 2), (4, 256, 512, 2), (2, 512, 512, 1)])
def test_conv_implicit_equals_column_matrix_path_at_baseline_size(H, C, co, s):
    """BASELINE cfg2 sizes (per-GPU batch 1024: ResNet-18 layers 1-4 at 64 x 64 input): the implicit-GEMM convolution and
    the im2col + GEMM + col2im path of the same library agree in y, dx and dw (same products, different operand plumbing)."""
    from multimodal_supernovae_amd import functional as F_
    g = torch.Generator().manual_seed(H * C + co)
    x = torch.randn(1024, H, H, C, generator=g).cuda()
    w = (torch.randn(co, C, 3, 3, generator=g) * 0.05).cuda()
    cot = torch.randn(1024, (H + 2 - 3) // s + 1, (H + 2 - 3) // s + 1, co, generator=g).cuda()
    res = []
    try:
        for implicit in (True, False):
            F_.CONV_IMPLICIT = implicit
            xg, wg = x.clone().requires_grad_(), w.clone().requires_grad_()
            y = F_.conv_cl(xg, wg, None, (s, s), (1, 1))
            y.backward(cot)
            res.append((y.detach(), xg.grad, wg.grad))
    finally:
        F_.CONV_IMPLICIT = True
    for a, b, what in zip(res[0], res[1], ("y", "dx", "dw")):
        close(a, b, what, rtol=2e-4)


@pytest.mark.parametrize("C", [5, 8, 64])       # scalar kernel, 4-channels-per-thread kernels
def test_maxpool_channels_last_matches_torch(C):
    from multimodal_supernovae_amd import functional as F_
    x = torch.randn(2, C, 9, 11, generator=torch.Generator().manual_seed(1))
    xr = x.clone().requires_grad_()
    ref = F.max_pool2d(xr, 3, 2, 1)
    cot = torch.randn(ref.shape, generator=torch.Generator().manual_seed(2))
    ref.backward(cot)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
    y = F_.maxpool_cl(xg, 3, 2, 1)
    close(y.detach().cpu().permute(0, 3, 1, 2), ref.detach(), "y")
    y.backward(cot.permute(0, 2, 3, 1).contiguous().cuda())
    close(xg.grad.cpu().permute(0, 3, 1, 2), xr.grad, "dx")


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_resnet18(mode):
    from multimodal_supernovae_amd.encoders import ResNet18
    from oracle.build_defined import resnet18
    torch.manual_seed(3)
    m = ResNet18(n_out=8)
    with torch.no_grad():
        for k, b in m.named_buffers():
            if k.endswith("running_var"):
                b.copy_(torch.rand_like(b) + 0.5)
            elif k.endswith("running_mean"):
                b.copy_(torch.randn_like(b) * 0.1)
    trainable = {k for k, _ in m.named_parameters()}
    P = {k: v.clone().requires_grad_(k in trainable) for k, v in m.state_dict().items()}
    x = torch.rand(6, 3, 64, 64)
    cot = torch.randn(6, 8)
    ref = resnet18(P, "", x, training=(mode == "train"))
    (ref * cot).sum().backward()
    m.cuda().train(mode == "train")
    y = m(x.cuda())
    close(y.detach().cpu(), ref.detach(), "y", rtol=2e-3)
    y.backward(cot.cuda())
    for k, p in m.named_parameters():
        close(p.grad.cpu(), P[k].grad, "grad " + k, rtol=5e-3)


def test_conv1d_encoder():
    from multimodal_supernovae_amd.encoders import Conv1dEncoder
    from oracle.build_defined import conv1d_encoder
    torch.manual_seed(5)
    m = Conv1dEncoder(n_out=8, widths=(16, 32), kernel_size=5, time_norm=100.0)
    P = {k: v.clone().requires_grad_() for k, v in m.state_dict().items()}
    B, T = 5, 50
    g = torch.Generator().manual_seed(6)
    x, t = torch.randn(B, T, 1, generator=g), torch.sort(torch.rand(B, T, generator=g) * 100, dim=1)[0]
    mask = torch.zeros(B, T, dtype=torch.bool)
    for b in range(B):
        mask[b, :int(torch.randint(5, T + 1, (1,), generator=g))] = True
    cot = torch.randn(B, 8, generator=g)
    ref = conv1d_encoder(P, "", x, t, mask, n_layers=2, time_norm=100.0)
    (ref * cot).sum().backward()
    m.cuda()
    y = m(x.cuda(), t.cuda(), mask.cuda())
    close(y.detach().cpu(), ref.detach(), "y")
    y.backward(cot.cuda())
    for k, p in m.named_parameters():
        close(p.grad.cpu(), P[k].grad, "grad " + k, rtol=2e-3)


def test_cfg2_and_cfg4_towers_fill_the_slots():
    """BASELINE cfg2 (ResNet-18 + 1-D CNN) and a cfg4-style 3-tower with the 1-D CNN on spectra: one training step."""
    from multimodal_supernovae_amd.encoders import Conv1dEncoder, ResNet18, VisionTransformer
    from multimodal_supernovae_amd.models_multimodal import LightCurveImageCLIP
    tk = dict(n_out=8, emb=16, heads=4, depth=1, dropout=0.0, time_norm=1e4, agg="mean")
    ck = dict(dim=8, depth=1, channels=3, kernel_size=5, patch_size=4, n_out=8, dropout_prob=0.0)
    B = 8
    img = torch.rand(B, 3, 32, 32).cuda()
    lc = (torch.randn(B, 20).cuda(), torch.rand(B, 20).cuda() * 100, torch.ones(B, 20, dtype=torch.bool).cuda())
    sp = (torch.randn(B, 64).cuda(), torch.rand(B, 64).cuda() * 6000 + 3000, torch.ones(B, 64, dtype=torch.bool).cuda())
    m2 = LightCurveImageCLIP(enc_dim=16, nband=2, transformer_kwargs=tk, conv_kwargs=ck,
                             combinations=["host_galaxy", "lightcurve"], loss="softmax")
    m2.image_encoder, m2.lightcurve_encoder = ResNet18(n_out=8), Conv1dEncoder(n_out=8, widths=(16, 16))
    m2.cuda().train()
    loss = m2.training_step((img, *lc, None, None, None, None, None), 0)
    loss.backward()
    assert torch.isfinite(loss.detach()) and all(p.grad is not None for p in m2.parameters())
    m4 = LightCurveImageCLIP(enc_dim=16, nband=2, transformer_kwargs=tk, transformer_spectral_kwargs=tk, conv_kwargs=ck,
                             combinations=["host_galaxy", "lightcurve", "spectral"], loss="softmax")
    m4.image_encoder = VisionTransformer(img_size=32, patch_size=8, emb=32, depth=1, heads=2, n_out=8)
    m4.spectral_encoder = Conv1dEncoder(n_out=8, widths=(16,), time_norm=9000.0)
    m4.cuda().train()
    loss = m4.training_step((img, *lc, *sp, None, None), 0)
    loss.backward()
    assert torch.isfinite(loss.detach()) and all(p.grad is not None for p in m4.parameters())

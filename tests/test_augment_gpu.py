"""Row f3: the on-device augmentation.  Series noise (ref src/dataloader.py:119-137, :215-238) is pinned by fixtures the
REAL NoisyDataLoader.__iter__ produced under a fixed seed (tools/gen_golden.py --only augment: inputs, errors, the Gaussian
field the loader drew, its output).  The image branch (:96-114: uniform noise + torchvision RandomRotation) cannot be run
in the build image (torchvision is absent) and is compared with the reference's formula evaluated in plain torch on the
same random fields; its rotation sense stays unpinned."""
import pytest
import torch

from conftest import Fixture, golden_names

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_names("augment_"))
def test_series_noise_against_reference_loader(name):
    """augment_series on the field the reference loader drew == the batch the reference loader yielded."""
    from multimodal_supernovae_amd.augment import augment_batch, augment_series
    fx = Fixture(name)
    level, nb = fx.cfg["noise_level_mag"], fx.cfg["batches"]
    seen = 0
    for bi in range(nb):
        for kind in ("lc", "sp"):
            if f"b{bi}.x_{kind}" not in fx.groups["in"]:
                continue
            x, err, field = (fx.groups["in"][f"b{bi}.{k}_{kind}"].cuda() for k in ("x", "err", "field"))
            out = augment_series(x, err, level, g=field)
            torch.testing.assert_close(out.cpu(), fx.out[f"b{bi}.x_{kind}"], rtol=1e-6, atol=1e-6)
            seen += 1
    assert seen >= nb
    if fx.cfg["combinations"] == ["spectral", "lightcurve"]:      # the 9-tuple wrapper leaves everything else untouched
        x_lc, x_sp = fx.groups["in"]["b0.x_lc"].cuda(), fx.groups["in"]["b0.x_sp"].cuda()
        batch = (None, x_lc, x_lc * 0, x_lc > 0, x_sp, x_sp * 0, x_sp > 0, None, None)
        aug = augment_batch(batch, 0.25, level, magerr=fx.groups["in"]["b0.err_lc"].cuda(), specerr=fx.groups["in"]["b0.err_sp"].cuda())
        assert aug[0] is None and aug[2] is batch[2] and aug[3] is batch[3] and aug[5] is batch[5] and aug[6] is batch[6]
        assert aug[1].shape == x_lc.shape and aug[4].shape == x_sp.shape and not torch.equal(aug[1], x_lc)


@pytest.mark.parametrize("B,C,S", [(5, 3, 8), (16, 3, 60), (3, 1, 7)])
def test_image_noise_and_quarter_turns(B, C, S):
    from multimodal_supernovae_amd.augment import augment_images
    g = torch.Generator().manual_seed(B + S)
    img, u = torch.rand(B, C, S, S, generator=g) * 3, torch.rand(B, C, S, S, generator=g)
    rot = torch.randint(0, 4, (B,), generator=g)
    level = 0.3
    noisy = img + (2 * u - 1) * (level * torch.std(img))                      # ref :96-101
    ref = torch.stack([torch.rot90(noisy[i], int(rot[i]), dims=(1, 2)) for i in range(B)])   # ref :104-114
    out = augment_images(img.cuda(), level, u=u.cuda(), rot=rot.cuda())
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)


def test_series_noise_and_batch_wrapper():
    from multimodal_supernovae_amd.augment import augment_batch, augment_series
    g = torch.Generator().manual_seed(3)
    x, gn, err = torch.randn(7, 50, generator=g), torch.randn(7, 50, generator=g), torch.rand(7, 50, generator=g)
    out = augment_series(x.cuda(), err.cuda(), 0.7, g=gn.cuda())
    torch.testing.assert_close(out.cpu(), x + gn * err * 0.7, rtol=1e-6, atol=1e-6)       # ref :123
    batch = (torch.rand(4, 3, 16, 16).cuda(), x[:4].cuda(), torch.rand(4, 50).cuda(), torch.ones(4, 50, dtype=torch.bool).cuda(),
             None, None, None, torch.rand(4).cuda(), None)
    aug = augment_batch(batch, 0.1, 1.0, magerr=err[:4].cuda())
    assert len(aug) == 9 and aug[0].shape == batch[0].shape and aug[4] is None
    assert not torch.equal(aug[0], batch[0]) and not torch.equal(aug[1], batch[1]) and aug[2] is batch[2]
    # the image noise is bounded by level * std and the rotation preserves the pixel multiset statistics
    assert abs(float(aug[0].mean() - batch[0].mean())) < 0.05

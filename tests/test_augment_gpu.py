"""Row f3: the on-device augmentation against the reference's formulas (src/dataloader.py:88-287) evaluated with
plain torch on the same random fields (the fields are inputs, so the comparison is exact up to fp32 rounding)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


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

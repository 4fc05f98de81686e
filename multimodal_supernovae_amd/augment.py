"""On-device batch augmentation = NoisyDataLoader.__iter__ of the reference (src/dataloader.py:88-287) applied to
a 9-tuple that is already on the GPU: uniform image noise scaled by the batch standard deviation, a random
multiple-of-90-degree rotation per image, Gaussian noise on magnitudes / spectra scaled by their errors.
The random fields come from torch's device generator (the reference's CPU RNG stream cannot be reproduced);
the arithmetic runs in libmsn_hip (msn_augment_images / msn_augment_series)."""
import torch

from . import ops
from ._lib import check, lib, ptr, stream_ptr


def augment_images(imgs, noise_level, u=None, rot=None, generator=None):
    """imgs: (B, C, S, S) on the GPU.  u: optional U[0,1) field, rot: optional (B,) quarter turns (int32)."""
    B, C, H, W = imgs.shape
    if H != W:
        raise ValueError("90-degree rotations keep the shape only for square images")
    imgs = ops._f32c(imgs.contiguous(), "imgs")
    if u is None:
        u = torch.rand(imgs.shape, device=imgs.device, generator=generator)
    if rot is None:
        rot = torch.randint(0, 4, (B,), device=imgs.device, generator=generator, dtype=torch.int32)
    out = torch.empty_like(imgs)
    L = lib()
    nb = L.msn_augment_workspace_bytes()
    ws = torch.empty(nb // 4 + 1, dtype=torch.float32, device=imgs.device)
    check(L.msn_augment_images(ptr(imgs), ptr(u.contiguous()), ptr(rot.to(torch.int32).contiguous()), B, C, H,
                               float(noise_level), ptr(out), ptr(ws), nb, stream_ptr()), "msn_augment_images")
    return out


def augment_series(x, err, noise_level, g=None, generator=None):
    x = ops._f32c(x.contiguous(), "x")
    if g is None:
        g = torch.randn(x.shape, device=x.device, generator=generator)
    out = torch.empty_like(x)
    check(lib().msn_augment_series(ptr(x), ptr(g.contiguous()), ptr(err.contiguous().float()), x.numel(), float(noise_level),
                                   ptr(out), stream_ptr()), "msn_augment_series")
    return out


def augment_batch(batch, noise_level_img, noise_level_mag, magerr=None, specerr=None, generator=None):
    """Apply the loader's augmentation to a 9-tuple (x_img, x_lc, t_lc, mask_lc, x_sp, t_sp, mask_sp, z, cls)."""
    x_img, x_lc, t_lc, mask_lc, x_sp, t_sp, mask_sp, z, cls = batch
    if x_img is not None:
        x_img = augment_images(x_img, noise_level_img, generator=generator)
    if x_lc is not None and magerr is not None:
        x_lc = augment_series(x_lc, magerr, noise_level_mag, generator=generator)
    if x_sp is not None and specerr is not None:
        x_sp = augment_series(x_sp, specerr, noise_level_mag, generator=generator)
    return (x_img, x_lc, t_lc, mask_lc, x_sp, t_sp, mask_sp, z, cls)

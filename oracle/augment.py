"""CPU restatement (test infrastructure) of the arithmetic of NoisyDataLoader.__iter__ (ref src/dataloader.py:88-287) with the
random fields as explicit inputs.  Pinned by tests/golden/augment_*.npz (series branches, generated from the real loader);
the image branch has no reference fixture (torchvision is absent from the build image)."""
import torch


def series_noise(x, err, field, noise_level_mag):
    """ref src/dataloader.py:123 / :132-134 / :231-236: x + N(0,1) * err * noise_level_mag"""
    return x + field * err * noise_level_mag


def image_noise_rot90(imgs, u, quarter_turns, noise_level_img):
    """ref src/dataloader.py:96-114: imgs + (2u - 1) * level * std(imgs over the whole batch), then a rotation by
    quarter_turns[i] * 90 degrees per image (torchvision's RandomRotation([a, a]) rotates counter-clockwise)."""
    noisy = imgs + (2 * u - 1) * (noise_level_img * torch.std(imgs))
    return torch.stack([torch.rot90(noisy[i], int(quarter_turns[i]), dims=(1, 2)) for i in range(imgs.shape[0])])

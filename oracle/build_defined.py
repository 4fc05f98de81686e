"""Oracle for the BUILD-DEFINED encoders (TEST INFRASTRUCTURE; parity unpinned by the reference).

ResNet-18 / ViT / 1-D CNN are named by BASELINE.json but do not exist in the reference (SURVEY.md
section 0), so there is nothing to import or pin against: this file is the specification, in plain
torch ops over a flat state_dict mapping, that the HIP implementation is tested against.
"""
import torch
import torch.nn.functional as F


def _ln(P, name, x, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], eps)


def vision_transformer(P, prefix, img, *, patch, heads, depth):
    """Pre-norm ViT with a class token (see multimodal_supernovae_amd.encoders.VisionTransformer)."""
    w = P[prefix + "patch_embed.proj.weight"]
    e = w.shape[0]
    x = F.conv2d(img, w, P[prefix + "patch_embed.proj.bias"], stride=patch)          # (B, e, gh, gw)
    B = x.shape[0]
    x = x.flatten(2).transpose(1, 2)                                                 # (B, hw, e)
    x = torch.cat([P[prefix + "cls_token"].expand(B, -1, -1), x], dim=1) + P[prefix + "pos_embed"]
    T = x.shape[1]
    hd = e // heads
    for i in range(depth):
        b = f"{prefix}blocks.{i}."
        h = _ln(P, b + "norm1", x)
        qkv = (h @ P[b + "attn.qkv.weight"].T + P[b + "attn.qkv.bias"]).view(B, T, 3, heads, hd)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        att = torch.softmax(torch.einsum("bihd,bjhd->bhij", q, k) / hd ** 0.5, dim=-1)
        a = torch.einsum("bhij,bjhd->bihd", att, v).reshape(B, T, e)
        x = x + a @ P[b + "attn.proj.weight"].T + P[b + "attn.proj.bias"]
        h = _ln(P, b + "norm2", x)
        h = F.gelu(h @ P[b + "mlp.fc1.weight"].T + P[b + "mlp.fc1.bias"])
        x = x + h @ P[b + "mlp.fc2.weight"].T + P[b + "mlp.fc2.bias"]
    cls = _ln(P, prefix + "norm", x[:, 0])
    return cls @ P[prefix + "head.weight"].T + P[prefix + "head.bias"]

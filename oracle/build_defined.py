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


def _bn2d(P, name, x, training, eps=1e-5):
    if training:
        return F.batch_norm(x, None, None, P[name + ".weight"], P[name + ".bias"], True, 0.1, eps)
    return F.batch_norm(x, P[name + ".running_mean"], P[name + ".running_var"], P[name + ".weight"], P[name + ".bias"],
                        False, 0.1, eps)


def resnet18(P, prefix, img, training=True):
    """torchvision-style ResNet-18 (see multimodal_supernovae_amd.encoders.ResNet18)."""
    x = F.conv2d(img, P[prefix + "conv1.weight"], None, stride=2, padding=3)
    x = F.max_pool2d(torch.relu(_bn2d(P, prefix + "bn1", x, training)), 3, 2, 1)
    for li in range(1, 5):
        for bi in range(2):
            b = f"{prefix}layer{li}.{bi}."
            stride = 2 if (li > 1 and bi == 0) else 1
            h = F.conv2d(x, P[b + "conv1.weight"], None, stride=stride, padding=1)
            h = torch.relu(_bn2d(P, b + "bn1", h, training))
            h = _bn2d(P, b + "bn2", F.conv2d(h, P[b + "conv2.weight"], None, padding=1), training)
            idt = x
            if b + "downsample.0.weight" in P:
                idt = _bn2d(P, b + "downsample.1", F.conv2d(x, P[b + "downsample.0.weight"], None, stride=stride), training)
            x = torch.relu(h + idt)
    x = x.mean(dim=(2, 3))
    return x @ P[prefix + "fc.weight"].T + P[prefix + "fc.bias"]


def conv1d_encoder(P, prefix, x, t, mask, *, n_layers, time_norm):
    """Build-defined 1-D CNN series encoder (see multimodal_supernovae_amd.encoders.Conv1dEncoder)."""
    m = mask.to(x.dtype)
    xs = x.reshape(t.shape)
    h = torch.stack([xs * m, t / time_norm * m, m, torch.zeros_like(m)], dim=1)          # (B, 4, T)
    for i in range(n_layers):
        w = P[f"{prefix}convs.{i}.weight"]
        h = torch.relu(F.conv1d(h, w, P[f"{prefix}convs.{i}.bias"], padding=w.shape[-1] // 2))
    pooled = (h * m[:, None, :]).sum(dim=2) / m.sum(dim=1)[:, None]
    return pooled @ P[prefix + "projection.weight"].T + P[prefix + "projection.bias"]

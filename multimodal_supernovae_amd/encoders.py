"""Build-defined encoders for the reference's encoder slots (NOT present in the reference, which
ships ConvMixer / TransformerWithTimeEmbeddings / MLP only -- SURVEY.md section 0).  BASELINE.json names
ViT-S/8, ViT-B/16, ResNet-18 and a 1-D CNN; their arithmetic is specified here and pinned by the
build's own CPU restatement (oracle/build_defined.py): "parity unpinned by the reference".

Slot contract (ref src/models_multimodal.py:275-293):  image_encoder: (B, C, H, W) -> (B, n_out).
"""
import torch
import torch.nn as nn

from . import functional as F_


class _Mlp(nn.Module):
    def __init__(self, emb, hidden):
        super().__init__()
        self.fc1 = nn.Linear(emb, hidden)
        self.fc2 = nn.Linear(hidden, emb)


class _Attn(nn.Module):
    def __init__(self, emb):
        super().__init__()
        self.qkv = nn.Linear(emb, 3 * emb)
        self.proj = nn.Linear(emb, emb)


class _Block(nn.Module):
    def __init__(self, emb, mlp_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(emb, eps=1e-6)
        self.attn = _Attn(emb)
        self.norm2 = nn.LayerNorm(emb, eps=1e-6)
        self.mlp = _Mlp(emb, mlp_ratio * emb)


class _PatchEmbed(nn.Module):
    def __init__(self, channels, emb, patch):
        super().__init__()
        self.proj = nn.Conv2d(channels, emb, kernel_size=patch, stride=patch)


class VisionTransformer(nn.Module):
    """Standard pre-norm ViT (timm-style parameter names): patch-embedding conv (with bias) ->
    [cls ; patches] + learned positions -> `depth` blocks {LN, MHA(qkv bias), +x, LN, MLP(GELU, x4), +x}
    -> final LN -> class token -> Linear(emb, n_out).  LayerNorm eps 1e-6, softmax scale 1/sqrt(head_dim).
    ViT-S/8: emb 384, heads 6, depth 12, patch 8 (T = 65 at 64x64);  ViT-B/16: 768 / 12 / 12 / 16."""

    def __init__(self, img_size=64, patch_size=8, channels=3, emb=384, depth=12, heads=6, mlp_ratio=4, n_out=32):
        super().__init__()
        assert emb % heads == 0
        self.patch_size, self.emb, self.heads, self.depth = patch_size, emb, heads, depth
        self.grid = img_size // patch_size
        self.num_tokens = 1 + self.grid * self.grid
        self.patch_embed = _PatchEmbed(channels, emb, patch_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, emb))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_tokens, emb))
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.blocks = nn.ModuleList([_Block(emb, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(emb, eps=1e-6)
        self.head = nn.Linear(emb, n_out)

    def forward(self, x):
        B = x.shape[0]
        T, e = self.num_tokens, self.emb
        if x.shape[2] // self.patch_size != self.grid or x.shape[3] // self.patch_size != self.grid:
            raise ValueError(f"image {tuple(x.shape)} does not match the {self.grid}x{self.grid} patch grid")
        patches = F_.patchify(x, self.patch_size)
        pe = self.patch_embed.proj
        emb = F_.linear(patches, pe.weight.view(e, -1), pe.bias)
        h = F_.vit_tokens(emb, self.cls_token, self.pos_embed, B, T)
        for blk in self.blocks:
            h = F_.pre_norm_block(h, self.heads, (blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight,
                                                  blk.attn.qkv.bias, blk.attn.proj.weight, blk.attn.proj.bias,
                                                  blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight,
                                                  blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias), eps=blk.norm1.eps)
        cls = F_.take_token(h, 0)
        cls = F_.layer_norm(cls, self.norm.weight, self.norm.bias, self.norm.eps)   # LN is per token: only the read-out needs it
        return F_.linear(cls, self.head.weight, self.head.bias)


def vit_s8(img_size=64, n_out=32, channels=3):
    return VisionTransformer(img_size=img_size, patch_size=8, channels=channels, emb=384, depth=12, heads=6, n_out=n_out)


def vit_b16(img_size=224, n_out=32, channels=3):
    return VisionTransformer(img_size=img_size, patch_size=16, channels=channels, emb=768, depth=12, heads=12, n_out=n_out)

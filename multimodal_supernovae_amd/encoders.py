"""Build-defined encoders for the reference's encoder slots (NOT present in the reference, which
ships ConvMixer / TransformerWithTimeEmbeddings / MLP only -- SURVEY.md section 0).  BASELINE.json names
ViT-S/8, ViT-B/16, ResNet-18 and a 1-D CNN; their arithmetic is specified here and pinned by the
build's own CPU restatement (oracle/build_defined.py): "parity unpinned by the reference".

Slot contract (ref src/models_multimodal.py:275-293):  image_encoder: (B, C, H, W) -> (B, n_out).
"""
import torch
import torch.nn as nn

from . import functional as F_
from .models_multimodal import MLP


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

    def __init__(self, img_size=64, patch_size=8, channels=3, emb=384, depth=12, heads=6, mlp_ratio=4, n_out=32,
                 gemm_precision=None):
        super().__init__()
        assert emb % heads == 0
        # None = the process default; "f32" = native fp32 MFMA; "bf16x6" = fp32-grade from three resident bf16 planes (six
        # bf16 MFMA products); "bf16x3p" / "bf16x3" = two planes / split in registers (three products, ~1e-5); "bf16" =
        # operands rounded to bf16 (BASELINE cfg5).  Attention, norms and the loss stay fp32.
        self.gemm_precision = gemm_precision
        self.cls_only_last_block = True      # False: evaluate every token of the last block (same result, more work)
        self.bf16_resident = True            # gemm_precision "bf16": bf16 activations in HBM + the 256-wide LDS-DMA GEMM
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
        if self.gemm_precision is None:
            return self._forward(x)
        from . import ops
        with ops.gemm_precision(self.gemm_precision):
            return self._forward(x)

    def _forward(self, x):
        B = x.shape[0]
        T, e = self.num_tokens, self.emb
        if x.shape[2] // self.patch_size != self.grid or x.shape[3] // self.patch_size != self.grid:
            raise ValueError(f"image {tuple(x.shape)} does not match the {self.grid}x{self.grid} patch grid")
        patches = F_.patchify(x, self.patch_size)
        pe = self.patch_embed.proj
        last = len(self.blocks) - 1
        dense_blocks = last if self.cls_only_last_block else last + 1
        # BASELINE cfg5: bf16-resident operands, 256 x 256 LDS-DMA tiles -- trunk, last block's token-sized products, patch embedding
        resident = (self.gemm_precision == "bf16" and self.bf16_resident and dense_blocks > 0 and e % 64 == 0
                    and all(b.norm1.eps == self.blocks[0].norm1.eps for b in self.blocks))
        emb = (F_.bf16_linear if resident else F_.linear)(patches, pe.weight.view(e, -1), pe.bias)
        h = F_.vit_tokens(emb, self.cls_token, self.pos_embed, B, T)
        params = [(blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.proj.weight,
                   blk.attn.proj.bias, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
                   blk.mlp.fc2.weight, blk.mlp.fc2.bias) for blk in self.blocks]
        from . import ops
        dense = len(params) - 1 if self.cls_only_last_block else len(params)       # blocks evaluated for every token
        first = 0
        if resident:
            # (functional._Bf16VitTrunk)
            h = F_.bf16_vit_trunk(h, self.heads, self.blocks[0].norm1.eps, params[:dense])
            first = dense
        elif (dense > 0 and F_.plane_path_ok(h) and all(b.norm1.eps == self.blocks[0].norm1.eps for b in self.blocks)):
            # "bf16x6" / "bf16x3p": fp32-grade products from resident bf16 planes (functional._PlaneVitTrunk)
            h = F_.plane_vit_trunk(h, self.heads, self.blocks[0].norm1.eps, params[:dense])
            first = dense
        for i in range(first, len(params)):
            if i == last and self.cls_only_last_block:
                # the head reads the class token only: the last block is evaluated for that row alone (identical output
                # and gradients; keys / values still come from every token)
                h = F_.pre_norm_last_block(h, self.heads, params[i], eps=self.blocks[i].norm1.eps, bf16_resident=resident)
            else:
                h = F_.pre_norm_block(h, self.heads, params[i], eps=self.blocks[i].norm1.eps)
        cls = h if (last >= 0 and self.cls_only_last_block) else F_.take_token(h, 0)
        cls = F_.layer_norm(cls, self.norm.weight, self.norm.bias, self.norm.eps)   # LN is per token: only the read-out needs it
        return F_.linear(cls, self.head.weight, self.head.bias)


def vit_s8(img_size=64, n_out=32, channels=3):
    return VisionTransformer(img_size=img_size, patch_size=8, channels=channels, emb=384, depth=12, heads=6, n_out=n_out)


def vit_b16(img_size=224, n_out=32, channels=3, gemm_precision="bf16"):
    """BASELINE cfg5: ViT-B/16 with its GEMMs on the bf16 matrix cores."""
    return VisionTransformer(img_size=img_size, patch_size=16, channels=channels, emb=768, depth=12, heads=12, n_out=n_out,
                             gemm_precision=gemm_precision)


# ------------------------------------------------------------------------------------------ ResNet-18
class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.stride = stride
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):  # x: (B, H, W, C) channels-last
        tr = self.training
        h = F_.conv_cl(x, self.conv1.weight, None, (self.stride, self.stride), (1, 1))
        h = F_.batchnorm_act(h, self.bn1, tr, relu=True)
        h = F_.conv_cl(h, self.conv2.weight, None, (1, 1), (1, 1))
        idt = x
        if self.downsample is not None:
            idt = F_.conv_cl(x, self.downsample[0].weight, None, (self.stride, self.stride), (0, 0))
            idt = F_.batchnorm_act(idt, self.downsample[1], tr)
        return F_.batchnorm_act(h, self.bn2, tr, residual=idt, relu=True)


class ResNet18(nn.Module):
    """torchvision-style ResNet-18 (same parameter names: conv1, bn1, layer{1..4}.{0,1}.*, fc) computing on
    channels-last tensors: 7x7/2 stem, 3x3/2 max pool, BasicBlock x [2, 2, 2, 2] (64, 128, 256, 512 channels),
    global average pool, Linear(512, n_out).  Convolutions = im2col + MFMA GEMM; BN(+ReLU, +skip) fused kernels."""

    def __init__(self, n_out=32, channels=3):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        widths, cin = [64, 128, 256, 512], 64
        for i, w in enumerate(widths):
            stride = 1 if i == 0 else 2
            setattr(self, f"layer{i + 1}", nn.Sequential(_BasicBlock(cin, w, stride), _BasicBlock(w, w, 1)))
            cin = w
        self.fc = nn.Linear(512, n_out)

    def forward(self, x):
        with F_.collect_batchnorm_counters():
            h = F_.to_channels_last(x)
            h = F_.conv_cl(h, self.conv1.weight, None, (2, 2), (3, 3))
            h = F_.batchnorm_act(h, self.bn1, self.training, relu=True)
            h = F_.maxpool_cl(h, 3, 2, 1)
            for i in range(1, 5):
                for blk in getattr(self, f"layer{i}"):
                    h = blk(h)
        B, H, W, C = h.shape
        ones = torch.ones((B, H * W), dtype=torch.uint8, device=h.device)
        pooled = F_.masked_pool(h.reshape(B, H * W, C), ones, "mean")
        return F_.linear(pooled, self.fc.weight, self.fc.bias)


# ------------------------------------------------------------------------------- 1-D CNN series encoder
class Conv1dEncoder(nn.Module):
    """Build-defined 1-D CNN for the light-curve / spectrum slots: (x (B,T,1), t (B,T), mask (B,T)) -> (B, n_out).
    Input channels (x*m, t/time_norm*m, m, 0); `len(widths)` x [Conv1d(k, padding=k//2) + ReLU]; masked mean over
    the valid positions; Linear(widths[-1], n_out).  Parameter names: convs.{i}.{weight,bias}, projection.*."""

    def __init__(self, n_out=32, widths=(64, 128, 128), kernel_size=5, time_norm=100.0):
        super().__init__()
        assert kernel_size % 2 == 1
        self.kernel_size, self.time_norm = kernel_size, float(time_norm)
        self.convs = nn.ModuleList()
        cin = 4
        for w in widths:
            self.convs.append(nn.Conv1d(cin, w, kernel_size, padding=kernel_size // 2))
            cin = w
        self.projection = nn.Linear(cin, n_out)

    def forward(self, x, t, mask=None):
        if mask is None:
            raise TypeError("Conv1dEncoder needs the (B, T) padding mask, like the reference's series encoders")
        from . import ops
        B, T = t.shape
        m = ops._mask_u8(mask)
        h = _SeriesFeatures.apply(x.reshape(B, T).float(), t.float(), m, 1.0 / self.time_norm).view(B, 1, T, 4)
        k = self.kernel_size
        for conv in self.convs:
            h = F_.conv_cl(h, conv.weight.unsqueeze(2), conv.bias, (1, 1), (0, k // 2), relu=True)
        pooled = F_.masked_pool(h.view(B, T, -1), m, "mean")
        return F_.linear(pooled, self.projection.weight, self.projection.bias)


class _SeriesFeatures(torch.autograd.Function):
    """Input featurisation; the inputs are data (no gradient)."""

    @staticmethod
    def forward(ctx, x, t, mask_u8, inv_norm):
        from . import ops
        return ops.series_features(x.contiguous(), t.contiguous(), mask_u8, inv_norm)

    @staticmethod
    def backward(ctx, d):
        return None, None, None, None


# ------------------------------------------------------------------- MLP in a series slot (BASELINE cfg1)
class SeriesMLP(MLP):
    """BASELINE.json configs[0]: the reference's `MLP` (ref src/models_multimodal.py:834-856; `num_layers` x
    (Linear, ReLU, Dropout) then Linear, state_dict keys layers.{0,3,6,...}) filling the light-curve slot
    `(x (B,T,1), t (B,T), mask (B,T)) -> (B, n_out)`: it acts on the flattened magnitudes (SURVEY.md section 8(d):
    "MLP(50->128->128->32) on flattened mags"); the time stamps and the mask are accepted for the slot signature
    and unused, as a plain MLP has no notion of either."""

    def __init__(self, seq_len, hidden_dim=128, n_out=32, num_layers=2, dropout=0.0):
        super().__init__(input_dim=seq_len, hidden_dim=hidden_dim, output_dim=n_out, num_layers=num_layers,
                         dropout=dropout)

    def forward(self, x, t=None, mask=None):
        flat = x.reshape(x.shape[0], -1).float()
        if flat.shape[1] != self.input_dim:
            raise ValueError(f"SeriesMLP was built for {self.input_dim} time steps, got {flat.shape[1]}")
        return super().forward(flat)

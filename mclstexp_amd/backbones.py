"""Image backbones (kernel K10): module trees + parameter names only.  EXECUTION: DenseNet-121 in bf16
runs entirely on the hand-written kernels scheduled by densenet_fused.py (``ImageEncoder.forward_fused`` /
``forward_eval_fused``), the ViT on vit_fused.py, the ResNets (a12) on resnet_fused.py (generic im2col + own-GEMM convolutions,
fp32 or bf16); the modules' own ``forward`` (plain torch.nn on PyTorch-ROCm) is what the A/B flag ``fused_backbone=False``
and CPU-side tooling use.

torchvision and timm are absent from the image, so the architectures are restated here in plain
``torch.nn`` with torchvision-/timm-compatible parameter names, which keeps reference checkpoints
(``image_encoder.model.0.denseblock1.denselayer1.conv1.weight`` ...) loadable.  The wrappers mirror
/root/reference/model.py:72-148: ``nn.Sequential(*children[:-1])`` then ``adaptive_avg_pool2d`` and
flatten -- for DenseNet that means pooling directly on ``norm5`` (no ReLU, model.py:82-84).

Backbone parity is unpinned (the reference does not vendor or pin torchvision/timm); see DESIGN.md.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import List, Sequence

import torch
import torch.nn.functional as F
from torch import nn


# ------------------------------------------------------------------ DenseNet-121 (torchvision layout)
class _DenseLayer(nn.Module):
    def __init__(self, c_in: int, growth: int, bn_size: int):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(c_in)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(c_in, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)

    def forward(self, feats: List[torch.Tensor]) -> torch.Tensor:
        x = torch.cat(feats, 1)
        x = self.conv1(self.relu1(self.norm1(x)))
        return self.conv2(self.relu2(self.norm2(x)))


class _DenseBlock(nn.ModuleDict):
    def __init__(self, n_layers: int, c_in: int, bn_size: int, growth: int):
        super().__init__()
        for i in range(n_layers):
            self["denselayer%d" % (i + 1)] = _DenseLayer(c_in + i * growth, growth, bn_size)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        feats = [x]
        for layer in self.values():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class _Transition(nn.Sequential):
    def __init__(self, c_in: int, c_out: int):
        super().__init__()
        self.norm = nn.BatchNorm2d(c_in)
        self.relu = nn.ReLU(inplace=True)
        self.conv = nn.Conv2d(c_in, c_out, 1, bias=False)
        self.pool = nn.AvgPool2d(2, 2)


def densenet121_features_module(growth: int = 32, block_config: Sequence[int] = (6, 12, 24, 16),
                                init_features: int = 64, bn_size: int = 4) -> nn.Sequential:
    """torchvision ``densenet121().features``, default torchvision init."""
    layers = OrderedDict()
    layers["conv0"] = nn.Conv2d(3, init_features, 7, stride=2, padding=3, bias=False)
    layers["norm0"] = nn.BatchNorm2d(init_features)
    layers["relu0"] = nn.ReLU(inplace=True)
    layers["pool0"] = nn.MaxPool2d(3, stride=2, padding=1)
    c = init_features
    for i, n in enumerate(block_config):
        layers["denseblock%d" % (i + 1)] = _DenseBlock(n, c, bn_size, growth)
        c += n * growth
        if i != len(block_config) - 1:
            layers["transition%d" % (i + 1)] = _Transition(c, c // 2)
            c //= 2
    layers["norm5"] = nn.BatchNorm2d(c)
    feats = nn.Sequential(layers)
    for m in feats.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    return feats


# ------------------------------------------------------------------ ResNets (torchvision layout)
class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, c_in, c, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(c_in, c, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(c)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(c)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, c_in, c, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(c_in, c, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(c)
        self.conv2 = nn.Conv2d(c, c, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(c)
        self.conv3 = nn.Conv2d(c, c * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(c * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


def _resnet_children(block, layers: Sequence[int]) -> List[nn.Module]:
    """children of torchvision ``resnetXX()`` minus ``fc``: conv1, bn1, relu, maxpool, layer1-4, avgpool."""
    mods: List[nn.Module] = [nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                             nn.MaxPool2d(3, 2, 1)]
    c_in = 64
    for i, n in enumerate(layers):
        c, stride = 64 * 2 ** i, (1 if i == 0 else 2)
        blocks = []
        for j in range(n):
            s = stride if j == 0 else 1
            ds = None
            if s != 1 or c_in != c * block.expansion:
                ds = nn.Sequential(nn.Conv2d(c_in, c * block.expansion, 1, s, bias=False),
                                   nn.BatchNorm2d(c * block.expansion))
            blocks.append(block(c_in, c, s, ds))
            c_in = c * block.expansion
        mods.append(nn.Sequential(*blocks))
    mods.append(nn.AdaptiveAvgPool2d((1, 1)))
    for m in mods:
        for mm in m.modules():
            if isinstance(mm, nn.Conv2d):
                nn.init.kaiming_normal_(mm.weight, mode="fan_out", nonlinearity="relu")
    return mods


class _PooledSequential(nn.Module):
    """``self.model = Sequential(children[:-1])`` + adaptive_avg_pool2d + flatten (model.py:81-85)."""

    def __init__(self, children: List[nn.Module]):
        super().__init__()
        self.model = nn.Sequential(*children)
        for p in self.model.parameters():
            p.requires_grad = True

    def forward(self, x):
        x = self.model(x)
        x = F.adaptive_avg_pool2d(x, (1, 1))
        return x.view(x.size(0), -1)

    def forward_fused(self, x, act_dtype=torch.bfloat16):
        """ResNet encoders on this library's kernels (resnet_fused.py): train-mode BatchNorm when ``self.training``, the
        running statistics otherwise.  (ImageEncoder overrides this with the DenseNet execution.)"""
        from .resnet_fused import resnet_features
        return resnet_features(self.model, x, act_dtype, training=self.training)


class ImageEncoder(_PooledSequential):
    """DenseNet-121, model.py:72-85.  Output (B, 1024).

    ``forward`` is the plain torch module path (MIOpen BatchNorm, torch.cat).  ``forward_fused`` runs the
    same parameters through the concat-free / stats-caching execution of ``densenet_fused`` (hand-written
    BN+ReLU kernels); train mode on the GPU only."""
    out_dim = 1024

    def __init__(self):
        super().__init__([densenet121_features_module()])

    def forward_fused(self, x, act_dtype=torch.bfloat16, cuts=None, cut_blocks=()):
        from .densenet_fused import densenet_features_fused
        return densenet_features_fused(self.model[0], x, act_dtype, pooled=True, cuts=cuts, cut_blocks=cut_blocks)

    def forward_eval_fused(self, x, act_dtype=torch.bfloat16):
        """Eval-mode (running statistics) forward on the fused kernels: the inference path of evel_her2st.py:50."""
        from .densenet_fused import densenet_features_eval
        return densenet_features_eval(self.model[0], x, act_dtype, pooled=True)


class ImageEncoder_Resnet(_PooledSequential):
    """ResNet-50, model.py:88-101.  Output (B, 2048)."""
    out_dim = 2048

    def __init__(self):
        super().__init__(_resnet_children(_Bottleneck, (3, 4, 6, 3)))


class ImageEncdoer_res18(_PooledSequential):
    """ResNet-18, model.py:119-132 (reference's spelling kept).  Output (B, 512)."""
    out_dim = 512

    def __init__(self):
        super().__init__(_resnet_children(_BasicBlock, (2, 2, 2, 2)))


class ImageEncdoer_res101(_PooledSequential):
    """ResNet-101, model.py:135-148 (reference's spelling kept).  Output (B, 2048)."""
    out_dim = 2048

    def __init__(self):
        super().__init__(_resnet_children(_Bottleneck, (3, 4, 23, 3)))


# ------------------------------------------------------------------ ViT (timm layout)
class _ViTAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        x = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        return self.proj(x.transpose(1, 2).reshape(B, N, C))


class _ViTMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _ViTBlock(nn.Module):
    def __init__(self, dim, heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _ViTAttention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _ViTMlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _PatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, patch, patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class VisionTransformer(nn.Module):
    """timm ``VisionTransformer(num_classes=0, global_pool='avg')`` layout: cls token + learned
    positions, pre-LN blocks, mean over patch tokens, ``fc_norm`` (final ``norm`` is Identity when
    global_pool='avg' in current timm)."""

    def __init__(self, img_size=224, patch=32, dim=768, depth=12, heads=12):
        super().__init__()
        self.patch_embed = _PatchEmbed(patch, dim)
        n = (img_size // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.randn(1, n + 1, dim) * 0.02)
        self.blocks = nn.Sequential(*[_ViTBlock(dim, heads) for _ in range(depth)])
        self.norm = nn.Identity()
        self.fc_norm = nn.LayerNorm(dim, eps=1e-6)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        x = self.norm(self.blocks(x))
        return self.fc_norm(x[:, 1:].mean(dim=1))


_VIT_CFG = {
    "vit_base_patch32_224": dict(img_size=224, patch=32, dim=768, depth=12, heads=12),
    "vit_base_patch16_224": dict(img_size=224, patch=16, dim=768, depth=12, heads=12),
}


class ImageEncoder_VIT(nn.Module):
    """model.py:104-116: ``timm.create_model(model_name, pretrained, num_classes=0, global_pool='avg')``.
    Pretrained weights need network access and are not available here (random init)."""
    out_dim = 768

    def __init__(self, model_name="vit_base_patch32_224", pretrained=True, trainable=True):
        super().__init__()
        if model_name not in _VIT_CFG:
            raise ValueError(f"unknown ViT '{model_name}' (have {sorted(_VIT_CFG)})")
        self.model = VisionTransformer(**_VIT_CFG[model_name])
        for p in self.model.parameters():
            p.requires_grad = trainable

    def forward(self, x):
        return self.model(x)


class ImageEncoder_VIT16(ImageEncoder_VIT):
    """Extension (not a reference selector value): ViT-B/16, the model BASELINE.json configs[2] names; the reference's
    'vit' is timm's vit_base_patch32_224 (model.py:106)."""

    def __init__(self):
        super().__init__("vit_base_patch16_224")


ENCODERS = {
    "resnet50": ImageEncoder_Resnet,
    "densenet121": ImageEncoder,
    "vit": ImageEncoder_VIT,
    "vit_b16": ImageEncoder_VIT16,
    "res18": ImageEncdoer_res18,
    "res101": ImageEncdoer_res101,
}

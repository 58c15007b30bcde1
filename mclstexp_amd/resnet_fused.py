"""ResNet image encoders (/root/reference/model.py:88-101 resnet50, 119-132 resnet18, 135-148 resnet101:
``nn.Sequential(*list(torchvision.models.resnetXX().children())[:-1])`` followed by adaptive_avg_pool2d + flatten) executed on
this library's kernels -- same module tree and parameters as ``backbones._resnet_children`` (torchvision layout: reference
checkpoints load), only the execution differs:

  * stem 7x7/2 convolution: the hand-written conv0 kernel of the DenseNet stem in bf16 (same 3 -> 64 shape; its epilogue also
    yields bn1's batch statistics), the generic path in fp32;
  * every other convolution (3x3 and 1x1, stride 1 and 2, 64..2048 channels): im2col + own GEMM (conv_generic.ConvFn:
    ``mcl_gemm_bf16`` for bf16 activations, the exact-fp32 MFMA ``mcl_gemm`` for fp32); 1x1 stride-1 convolutions are GEMMs on
    the activation as it lies in memory;
  * BatchNorm (train mode: batch statistics, running-statistics bookkeeping in one launch for all layers) + ReLU:
    csrc/bnrelu.hip (the DenseNet path's kernels, any channel count, fp32 or bf16);
  * MaxPool2d(3, 2, 1), the residual add + ReLU and the global average pool: csrc/pool_generic.hip.

No MIOpen / ATen convolution, BatchNorm or pooling call.  Activations are channels-last, bf16 (``backbone_dtype=bf16``) or
fp32 (``backbone_dtype=None``).  Forward and backward are hand-written autograd Functions throughout.
"""
from __future__ import annotations

import torch
from torch import nn

from . import conv_generic as cg
from . import densenet_fused as dn

Tensor = torch.Tensor
CL = torch.channels_last


def _bn(x: Tensor, bn: nn.BatchNorm2d, relu: bool, rec, training: bool) -> Tensor:
    if training:
        return dn._bn_train(x, bn, relu, rec)
    # eval: the affine map of the running statistics
    x = x.contiguous(memory_format=CL)
    rstd = dn.eval_rstd([bn])[0]
    out = torch.empty_like(x, memory_format=CL)
    dn.bn_act_fwd(x, bn.weight, bn.bias, bn.running_mean, rstd, relu, out)
    return out


def _block(blk: nn.Module, x: Tensor, rec, training: bool) -> Tensor:
    x, idt = cg.fork2(x)                      # two consumers: the block and its shortcut (their gradients meet in an own kernel)
    if blk.downsample is not None:
        ds_conv, ds_bn = blk.downsample[0], blk.downsample[1]
        idt = _bn(cg.conv2d(idt, ds_conv.weight, ds_conv.stride[0], ds_conv.padding[0]), ds_bn, False, rec, training)
    if hasattr(blk, "conv3"):                                      # Bottleneck: 1x1 -> 3x3 (stride) -> 1x1
        out = _bn(cg.conv2d(x, blk.conv1.weight, 1, 0), blk.bn1, True, rec, training)
        out = _bn(cg.conv2d(out, blk.conv2.weight, blk.conv2.stride[0], 1), blk.bn2, True, rec, training)
        out = _bn(cg.conv2d(out, blk.conv3.weight, 1, 0), blk.bn3, False, rec, training)
    else:                                                          # BasicBlock: 3x3 (stride) -> 3x3
        out = _bn(cg.conv2d(x, blk.conv1.weight, blk.conv1.stride[0], 1), blk.bn1, True, rec, training)
        out = _bn(cg.conv2d(out, blk.conv2.weight, 1, 1), blk.bn2, False, rec, training)
    return cg.add_relu(out, idt)


def resnet_features(seq: nn.Sequential, x: Tensor, act_dtype: torch.dtype = torch.bfloat16, training: bool = True) -> Tensor:
    """``seq`` = Sequential(conv1, bn1, relu, maxpool, layer1..4, avgpool) (backbones._resnet_children).  Returns the (B, C)
    fp32 features after the global average pool + flatten (model.py:98-99)."""
    if not x.is_cuda:
        raise RuntimeError("resnet_features: input is on the CPU; the fused backbone path is GPU-only")
    mods = list(seq.children())
    conv1, bn1 = mods[0], mods[1]
    rec = dn._RunningStats()
    if act_dtype == torch.bfloat16:
        x = dn.image_to_act(x, act_dtype)
    else:
        x = x.to(dtype=act_dtype)
        if not x.is_contiguous(memory_format=CL):
            x = x.contiguous(memory_format=CL)
    if training and dn._conv0_ok(x, conv1):
        mean0, var0, rstd0 = (torch.empty(64, device=x.device, dtype=torch.float32) for _ in range(3))
        y = dn.Conv0Fn.apply(x, conv1.weight, (bn1.eps, (mean0, var0, rstd0)))
        rec.add(bn1, mean0, var0, y.numel() // 64)
        y = dn.BNActFn.apply(y, bn1.weight, bn1.bias, mean0, rstd0, True)
    elif dn._conv0_ok(x, conv1):                                   # eval: the same stem kernel without the statistics
        y = _bn(dn.conv0_fwd(x, dn._weight(conv1.weight, act_dtype), bn1.eps, None), bn1, True, rec, training)
    else:
        if act_dtype == torch.bfloat16:
            raise RuntimeError(f"resnet_features: bf16 stem convolution on an input of shape {tuple(x.shape)} is outside the "
                               "stem kernel's range (H % 4 == 0, W % 8 == 0, W <= 256) and the bf16 GEMM needs K % 8 == 0; "
                               "use backbone_dtype=None (fp32) for this input size")
        y = _bn(cg.conv2d(x, conv1.weight, conv1.stride[0], conv1.padding[0]), bn1, True, rec, training)
    y = cg.max_pool_3s2(y)
    for layer in mods[4:]:
        if isinstance(layer, nn.Sequential):
            for blk in layer:
                y = _block(blk, y, rec, training)
    out = cg.global_avg_pool(y)
    if training:
        rec.flush()
    return out

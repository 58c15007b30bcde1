#!/usr/bin/env python3
"""Worst / median parameter-gradient deviation from an fp64 run of the same DenseNet-121 (train mode, random init, random
upstream gradient) for the stock bf16-autocast modules and the fused bf16 path, by batch size:
    python tools/diag_accuracy_batch.py B [HW]          (env switches select the fused path's variants)"""
import copy
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mclstexp_amd import backbones, densenet_fused as dn

B = int(sys.argv[1]); HW = int(sys.argv[2]) if len(sys.argv) > 2 else 224
torch.manual_seed(0)
base = backbones.ImageEncoder()
ref64 = copy.deepcopy(base).double().cuda().train()
ref = copy.deepcopy(base).cuda().train()
fus = copy.deepcopy(base).cuda().to(memory_format=torch.channels_last).train()
g = torch.Generator().manual_seed(3)
x = torch.rand(B, 3, HW, HW, generator=g).cuda()
dy = (torch.rand(B, 1024, generator=g) - 0.5).cuda()
y64 = ref64(x.double()); y64.backward(dy.double())
with torch.autocast("cuda", dtype=torch.bfloat16):
    y_ref = ref(x.contiguous(memory_format=torch.channels_last)).float()
y_ref.backward(dy)
for p in fus.parameters():
    p.grad = torch.zeros_like(p)
y = fus.forward_fused(x, torch.bfloat16); y.backward(dy)
names, d_ref, d_fus = [], [], []
for (n, p64), (_, p), (_, q) in zip(ref64.named_parameters(), ref.named_parameters(), fus.named_parameters()):
    s_ = p64.grad.abs().max().item() + 1e-30
    names.append(n)
    d_ref.append((p.grad.double() - p64.grad).abs().max().item() / s_)
    d_fus.append((q.grad.double() - p64.grad).abs().max().item() / s_)
d_ref, d_fus = np.array(d_ref), np.array(d_fus)
i = int(d_fus.argmax())
print(f"B={B} {HW}px single_pass={os.environ.get('MCL_BN1_SINGLE_PASS', '1')}: median stock {np.median(d_ref):.3f} fused {np.median(d_fus):.3f}; "
      f"p90 stock {np.quantile(d_ref, 0.9):.3f} fused {np.quantile(d_fus, 0.9):.3f}; max stock {d_ref.max():.2f} fused {d_fus.max():.2f} "
      f"({names[i]}, stock there {d_ref[i]:.2f})")

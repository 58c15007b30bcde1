#!/usr/bin/env python3
"""Micro-benchmark of the stem convolution kernels (forward with statistics, weight gradient) at 128 x 224^2."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mclstexp_amd import _lib, densenet_fused as dn
from mclstexp_amd._lib import check
B, H, W = 128, 224, 224
x = torch.rand(B, 3, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(64, 3, 7, 7, device="cuda") * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
st = tuple(torch.empty(64, device="cuda") for _ in range(3))
dy = torch.randn(B, 64, H // 2, W // 2, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
dW = torch.zeros(64, 3, 7, 7, device="cuda").contiguous(memory_format=torch.channels_last)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tf = t(lambda: dn.conv0_fwd(x, w, 1e-5, st))
ws = torch.empty(_lib.lib().mcl_conv0_wrw_workspace_floats(B, H, W), device="cuda")
tw = t(lambda: check(_lib.lib().mcl_conv0_wrw(x.data_ptr(), B, H, W, dy.data_ptr(), ws.data_ptr(), dW.data_ptr(), 1, dn._stream()), "wrw"))
print(f"conv0 fwd+stats {tf:.1f} us ({(x.numel()*2 + B*64*H*W//4*2)/tf/1e6:.2f} TB/s)   wrw {tw:.1f} us")

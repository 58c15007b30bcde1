#!/usr/bin/env python3
"""Run-to-run determinism of the fused DenseNet forward kernels (diagnostic)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mclstexp_amd.backbones import ImageEncoder
from mclstexp_amd import densenet_fused as dn

torch.manual_seed(0)
enc = ImageEncoder().to("cuda").eval()
x = torch.rand(2, 3, 224, 224, device="cuda")
with torch.no_grad():
    ys = [enc.forward_eval_fused(x) for _ in range(4)]
print("pairs:", [[bool(torch.equal(a, b)) for b in ys] for a in ys])
import torch.nn.functional as F
f = enc.model[0]
xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w0 = f.conv0.weight.to(dtype=torch.bfloat16, memory_format=torch.channels_last)
c0 = [F.conv2d(xb, w0, stride=2, padding=3) for _ in range(5)]
print("conv0 repeat:", [bool(torch.equal(c0[0], c)) for c in c0[1:]])
mp = [dn.max_pool_3s2(c0[0].contiguous(memory_format=torch.channels_last)) for _ in range(5)]
print("maxpool repeat:", [bool(torch.equal(mp[0], c)) for c in mp[1:]])
for (C, H) in [(256, 56), (512, 28), (1024, 14)]:
    a = torch.randn(2, C, H, H, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(C // 2, C, 1, 1, device="cuda") * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    t = [F.conv2d(a, wt) for _ in range(5)]
    print("transition conv", C, H, [bool(torch.equal(t[0], c)) for c in t[1:]])
    ap = [dn.avg_pool_2(t[0].contiguous(memory_format=torch.channels_last)) for _ in range(5)]
    print("avgpool", [bool(torch.equal(ap[0], c)) for c in ap[1:]])
    bn = torch.nn.BatchNorm2d(C).to("cuda").eval()
    rs = torch.rsqrt(bn.running_var + 1e-5)
    outs = []
    for _ in range(5):
        o = torch.empty_like(a); dn.bn_act_fwd(a, bn.weight, bn.bias, bn.running_mean, rs, True, o); outs.append(o)
    print("bn_act", [bool(torch.equal(outs[0], c)) for c in outs[1:]])
print("eval fused repeat equal:", [bool(torch.equal(ys[0], y)) for y in ys[1:]], (ys[0] - ys[1]).abs().max().item(), ys[0].abs().max().item())

# kernel level
for (S_b, H, cin) in [(2, 56, 64), (2, 56, 224), (2, 28, 256), (2, 14, 512), (2, 7, 992), (128, 7, 512)]:
    B = S_b
    buf = (torch.randn(B, H, H, 1024, device="cuda")).to(torch.bfloat16).permute(0, 3, 1, 2)
    xs = buf[:, :cin]
    g1 = torch.rand(cin, device="cuda") + 0.5; b1 = torch.rand(cin, device="cuda") - 0.5
    mean = torch.rand(cin, device="cuda") - 0.5; rstd = torch.rand(cin, device="cuda") + 0.5
    w1 = (torch.randn(128, cin, 1, 1, device="cuda") * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    zs = [dn.dense_conv1x1_fwd(xs, g1, b1, mean, rstd, w1, 1e-5, None, None, None) for _ in range(6)]
    torch.cuda.synchronize()
    e1 = [bool(torch.equal(zs[0], z)) for z in zs[1:]]
    g2 = torch.rand(128, device="cuda") + 0.5; b2 = torch.rand(128, device="cuda") - 0.5
    m2 = torch.rand(128, device="cuda") - 0.5; r2 = torch.rand(128, device="cuda") + 0.5
    w2 = (torch.randn(32, 128, 3, 3, device="cuda") * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    outs = []
    for _ in range(6):
        o = torch.zeros(B, H, H, 1024, device="cuda", dtype=torch.bfloat16).permute(0, 3, 1, 2)
        dn.dense_conv3x3_fwd(zs[0], g2, b2, m2, r2, w2, o[:, cin:cin + 32], 1e-5, None, None, None)
        outs.append(o)
    torch.cuda.synchronize()
    e3 = [bool(torch.equal(outs[0], o)) for o in outs[1:]]
    print(f"B{B} H{H} cin{cin}: conv1x1 repeat-equal {e1}  conv3x3 repeat-equal {e3}", flush=True)

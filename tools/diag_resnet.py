#!/usr/bin/env python3
"""Layer-by-layer comparison of the fused ResNet execution (resnet_fused.py) with an fp64 run of the same module."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mclstexp_amd import backbones, conv_generic as cg, densenet_fused as dn, resnet_fused as rf

torch.manual_seed(0)
dt = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == "f32") else torch.bfloat16
enc = backbones.ImageEncdoer_res18().cuda().to(memory_format=torch.channels_last).train()
ref = copy.deepcopy(enc).double()
x = torch.rand(4, 3, 64, 64, device="cuda")
mods, rmods = list(enc.model.children()), list(ref.model.children())
rec = dn._RunningStats()
xa = x.to(dt).contiguous(memory_format=torch.channels_last)
xr = x.double() if dt == torch.float32 else x.to(dt).double()
def rel(a, b): return float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))
y = cg.conv2d(xa, mods[0].weight, 2, 3); yr = rmods[0](xr); print("conv1", rel(y, yr))
y = rf._bn(y, mods[1], True, rec, True); yr = torch.relu(rmods[1](yr)); print("bn1+relu", rel(y, yr))
y = cg.max_pool_3s2(y); yr = rmods[3](yr); print("maxpool", rel(y, yr))
for li in range(4, 8):
    for bi, (blk, rblk) in enumerate(zip(mods[li], rmods[li])):
        idt, ridt = y, yr
        if blk.downsample is not None:
            c = cg.conv2d(y, blk.downsample[0].weight, blk.downsample[0].stride[0], 0); rc = rblk.downsample[0](yr)
            print(f"  layer{li-3}.{bi} ds conv", rel(c, rc))
            idt = rf._bn(c, blk.downsample[1], False, rec, True); ridt = rblk.downsample[1](rc)
            print(f"  layer{li-3}.{bi} ds bn", rel(idt, ridt))
        o = cg.conv2d(y, blk.conv1.weight, blk.conv1.stride[0], 1); ro = rblk.conv1(yr); print(f"  layer{li-3}.{bi} conv1", rel(o, ro))
        o = rf._bn(o, blk.bn1, True, rec, True); ro = torch.relu(rblk.bn1(ro)); print(f"  layer{li-3}.{bi} bn1", rel(o, ro))
        o = cg.conv2d(o, blk.conv2.weight, 1, 1); ro = rblk.conv2(ro); print(f"  layer{li-3}.{bi} conv2", rel(o, ro))
        o = rf._bn(o, blk.bn2, False, rec, True); ro = rblk.bn2(ro); print(f"  layer{li-3}.{bi} bn2", rel(o, ro))
        y = cg.add_relu(o, idt); yr = torch.relu(ro + ridt); print(f"layer{li-3}.{bi} out", rel(y, yr))
g = cg.global_avg_pool(y); gr = yr.mean((2, 3)); print("gap", rel(g, gr))
full = enc.forward_fused(x, dt); fr = ref(xr); print("full forward_fused vs module", rel(full, fr))

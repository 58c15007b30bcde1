import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mclstexp_amd.backbones import ImageEncoder
torch.manual_seed(0)
enc = ImageEncoder()
with torch.no_grad():
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.2, 0.2)
g = torch.Generator().manual_seed(5)
x = torch.rand(8, 3, 96, 96, generator=g).cuda()
dy = (torch.rand(8, 1024, generator=g) - 0.5).cuda()
ref64 = copy.deepcopy(enc).double().cuda().train()
ref32 = copy.deepcopy(enc).cuda().train()
fus32 = copy.deepcopy(enc).cuda().train()
y64 = ref64(x.double()); y64.backward(dy.double())
y32 = ref32(x); y32.backward(dy)
yf = fus32.forward_fused(x, torch.float32); yf.backward(dy)
print("features: module32 vs 64 %.3e   fused32 vs 64 %.3e" % ((y32.double()-y64).abs().max().item(), (yf.double()-y64).abs().max().item()))
rows = []
for (n, p64), (_, p32), (_, pf) in zip(ref64.named_parameters(), ref32.named_parameters(), fus32.named_parameters()):
    s = p64.grad.abs().max().item() + 1e-30
    rows.append((n, (p32.grad.double()-p64.grad).abs().max().item()/s, (pf.grad.double()-p64.grad).abs().max().item()/s))
rows.sort(key=lambda r: -r[2])
for r in rows[:12]: print("%-55s module32 %.2e  fused32 %.2e" % r)
print("max module32 %.2e  max fused32 %.2e" % (max(r[1] for r in rows), max(r[2] for r in rows)))

#!/usr/bin/env python3
"""Diagnostic: direct param-grad accumulation vs autograd hand-over, which parameter deviates and how often."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mclstexp_amd import densenet_fused as dn, synth
from mclstexp_amd.model import mclSTExp_Attention
from mclstexp_amd.optim import FusedAdam
DEV = "cuda"
if "det" in sys.argv:
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
G = 171
torch.manual_seed(0)
m = mclSTExp_Attention("densenet121", 100.0, 1024, G, 256, 8, 64, 2, backbone_dtype=None, embedding_grad="rowsparse")
sd = m.state_dict(); sd.update(synth.make_params(G, 1024, seed=0)); m.load_state_dict(sd)
m.to(DEV).train()
if len(sys.argv) > 1 and sys.argv[1] == "nooverlap":
    m.overlap_branches = False
opt = FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3).attach_model(m)
batch = {k: v.to(DEV) for k, v in synth.make_batch(8, G, image_hw=96, seed=0).items()}
loss = m(batch); opt.zero_grad(); loss.backward(); opt.step()
loss = m(batch); opt.zero_grad(); loss.backward()
def run(direct):
    dn.DIRECT_PARAM_GRADS = direct
    loss = m(batch); opt.zero_grad(); loss.backward()
    dn.DIRECT_PARAM_GRADS = True
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if n.startswith("image_encoder")}
base = run(False)
for it in range(12):
    direct = (it % 2 == 0)
    g = run(direct)
    worst = sorted(((( g[n] - base[n]).abs().max() / (base[n].abs().max() + 1e-20)).item(), n) for n in g)[-3:]
    print(it, "direct" if direct else "autograd", [(f"{w:.2e}", n[-45:]) for w, n in worst], flush=True)

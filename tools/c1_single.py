#!/usr/bin/env python3
"""conv1x1 forward alone at one shape, a few launches (for rocprofv3 --pmc):  python tools/c1_single.py S C ld [n]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mclstexp_amd import _lib, densenet_fused as dn  # noqa: E402

S, C, ld = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 5
L = _lib.lib()
dev = "cuda"
g = torch.Generator().manual_seed(0)
xw = ((torch.rand(S, ld, generator=g) - 0.4) * 2).to(torch.bfloat16).to(dev)
W1 = ((torch.rand(128, C, generator=g) - 0.5) / 8).to(torch.bfloat16).to(dev)
gam, bet, mu, rs = ((torch.rand(1024, generator=g) + 0.5).to(dev) for _ in range(4))
zo = torch.empty(S, 128, device=dev, dtype=torch.bfloat16)
zm, zv, zr = (torch.zeros(128, device=dev) for _ in range(3))
ws = torch.empty(L.mcl_dense_conv1x1_workspace_floats(S), device=dev)
P = lambda t: t.data_ptr()
for _ in range(n):
    _lib.check(L.mcl_dense_conv1x1_fwd(P(xw), ld, S, C, P(gam), P(bet), P(mu), P(rs), P(W1), P(zo), 128, P(ws), 1e-5, P(zm),
                                       P(zv), P(zr), dn._stream()), "c1")
torch.cuda.synchronize()

#!/usr/bin/env python3
"""Runs the fused InfoNCE lse + grad kernels a few times at one size (target for rocprofv3 --pmc / --kernel-trace).

    python3 tools/prof_infonce.py [B] [reps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mclstexp_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
_lib.lib()
P = 256
g = torch.Generator(device="cuda").manual_seed(B)
es = torch.nn.functional.layer_norm(torch.randn(B, P, device="cuda", generator=g), (P,))
ei = torch.nn.functional.layer_norm(torch.randn(B, P, device="cuda", generator=g) + 0.3 * es, (P,))
es16, ei16 = ops.cast_bf16(es), ops.cast_bf16(ei)
for _ in range(reps):
    rl, _d = ops.infonce_fused_lse(es16, ei16, 1.0)
    cl, _d = ops.infonce_fused_lse(ei16, es16, 1.0)
    d = ops.infonce_fused_grad(es16, ei16, 1.0, rl, cl, 1.0 / (2 * B))
torch.cuda.synchronize()
print("done", float(d.abs().max()))

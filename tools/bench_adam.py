#!/usr/bin/env python3
"""Micro-benchmark of the position-table Adam kernel (24 B/element, HBM-bound): python tools/bench_adam.py [G]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mclstexp_amd import _lib, ops
L = _lib.lib()
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N = 65536
p = torch.randn(N, G, device="cuda"); m = torch.zeros_like(p); v = torch.zeros_like(p)
slot = torch.full((N,), -1, device="cuda", dtype=torch.int32)
rows = torch.randn(128, G, device="cuda")
slot[:128] = torch.arange(128, device="cuda", dtype=torch.int32)
def f():
    _lib.check(L.mcl_adam_table_step(p.data_ptr(), m.data_ptr(), v.data_ptr(), N, G, slot.data_ptr(), rows.data_ptr(), G,
                                     1e-4, 0.9, 0.999, 1e-8, 1e-3, 0.1, 0.001, ops._stream()))
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"adam_table G={G}: {ms*1e3:.1f} us  {24.0*N*G/ms/1e6:.0f} GB/s")

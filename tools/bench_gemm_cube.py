#!/usr/bin/env python3
"""mcl_gemm_bf16 on square problems (the sizes the CDNA4 guide quotes its 256x256 template on), NT layout, random operands."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mclstexp_amd import vit_fused as vf  # noqa: E402

BF = torch.bfloat16


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


cases = [(4096, 4096, 4096), (8192, 8192, 8192), (8192, 8192, 768), (50432, 2304, 768), (50432, 2304, 4096)]
if len(sys.argv) > 3:
    cases = [tuple(int(v) for v in sys.argv[1:4])]
for (m, n, k) in cases:
    A = (torch.rand((m, k), device="cuda") * 2 - 1).to(BF)
    B = (torch.rand((n, k), device="cuda") * 2 - 1).to(BF)
    C = torch.empty((m, n), device="cuda", dtype=BF)
    t = timeit(lambda: vf.gemm(A, B, C, m, n, k, k, k, n))
    tt = timeit(lambda: torch.mm(A, B.t()))
    print(json.dumps({"M": m, "N": n, "K": k, "ms": round(t * 1e3, 3), "TFs": round(2.0 * m * n * k / t / 1e12, 1),
                      "hipblaslt_TFs": round(2.0 * m * n * k / tt / 1e12, 1)}), flush=True)

#!/usr/bin/env python3
"""Input pipeline micro-benchmark: 128 patches of 224 x 224 from a 20000 x 20000 slide resident in HBM."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mclstexp_amd import input_pipeline as ip
slide = torch.randint(0, 256, (20000, 20000, 3), dtype=torch.uint8, device="cuda")
rng = np.random.default_rng(0)
c = np.stack([rng.integers(200, 19800, 128), rng.integers(200, 19800, 128)], 1)
k = rng.integers(0, 4, 128)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cd = torch.as_tensor(c, device="cuda", dtype=torch.int32)
print(f"fp32 NCHW (the reference's tensor): {t(lambda: ip.extract_patches(slide, cd, 112, rot_k=k)):.1f} us; "
      f"bf16 NHWC (backbone input): {t(lambda: ip.extract_patches(slide, cd, 112, rot_k=k, layout='nhwc_bf16')):.1f} us "
      f"(the reference: 128 PIL crops + ToTensor on the host, then a 77 MB H2D copy)")
x = torch.rand(4096, 1000, device="cuda") * (torch.rand(4096, 1000, device="cuda") < 0.3)
print(f"log_library_size_normalize 4096 x 1000: {t(lambda: ip.log_library_size_normalize(x)):.1f} us")

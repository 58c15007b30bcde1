#!/usr/bin/env python3
"""Eval-mode DenseNet-121 image encoder (the inference half of SURVEY §8 f1, evel_her2st.py:50): fused kernels with
running statistics vs the stock PyTorch-ROCm module under bf16 autocast.  One JSON line per batch size.

    python tools/bench_inference.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mclstexp_amd.backbones import ImageEncoder  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    torch.manual_seed(0)
    enc = ImageEncoder().to("cuda").eval()
    for b in (32, 128, 256):
        x = torch.rand(b, 3, 224, 224, device="cuda")
        xcl = x.contiguous(memory_format=torch.channels_last)

        def stock():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                return enc(xcl)

        def fused():
            with torch.no_grad():
                return enc.forward_eval_fused(x)

        t_s, t_f = timeit(stock), timeit(fused)
        print(json.dumps({"batch": b, "image": 224, "stock_bf16_ms": round(t_s, 3), "fused_ms": round(t_f, 3),
                          "stock_img_per_s": round(b / t_s * 1e3), "fused_img_per_s": round(b / t_f * 1e3),
                          "speedup": round(t_s / t_f, 2)}), flush=True)


if __name__ == "__main__":
    main()

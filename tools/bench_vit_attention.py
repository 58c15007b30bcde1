#!/usr/bin/env python3
"""Micro-benchmark of the fused ViT attention core (csrc/vit_attention.hip) at the ViT-B/16 and B/32 batch-256 shapes: us per launch
and GB/s against the bytes each launch has to move (forward: qkv in, o out; backward: qkv, o, do in, dqkv out)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mclstexp_amd import _lib  # noqa: E402
from mclstexp_amd._lib import check  # noqa: E402

BF = torch.bfloat16
L = _lib.lib()


def timeit(fn, n=20):
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


for (B, T, heads) in [(256, 197, 12), (256, 50, 12)]:
    D = heads * 64
    qkv = torch.randn((B, T, 3 * D), device="cuda").to(BF)
    do = torch.randn((B, T, D), device="cuda").to(BF)
    o = torch.empty((B, T, D), device="cuda", dtype=BF)
    lse = torch.empty((B * heads, T), device="cuda")
    dsum = torch.empty((B * heads, T), device="cuda")
    dqkv = torch.empty_like(qkv)
    st = torch.cuda.current_stream().cuda_stream
    tf = timeit(lambda: check(L.mcl_vit_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, T, heads, 0.125, st), "fwd"))
    tb = timeit(lambda: check(L.mcl_vit_attn_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), dsum.data_ptr(),
                                                 dqkv.data_ptr(), B, T, heads, 0.125, st), "bwd"))
    bytes_f = (qkv.numel() + o.numel()) * 2
    bytes_b = (2 * qkv.numel() + 3 * o.numel() + qkv.numel()) * 2      # dq kernel: qkv, o, do; dkv kernel: qkv, do; dqkv out
    flop_f = 4.0 * B * heads * T * T * 64
    print(json.dumps({"B": B, "T": T, "heads": heads, "fwd_us": round(tf * 1e6, 1), "fwd_GBps": round(bytes_f / tf / 1e9),
                      "bwd_us": round(tb * 1e6, 1), "bwd_GBps": round(bytes_b / tb / 1e9),
                      "fwd_TFs": round(flop_f / tf / 1e12, 1), "bwd_TFs": round(3.5 * flop_f / tb / 1e12, 1)}), flush=True)

#!/usr/bin/env python3
"""The projection head (model.py:151-168) forward + backward as a replayed HIP graph: the single-launch kernels
(csrc/proj_head.hip + mcl_gemm_group) against the separate epilogue-fused launches (MCL_FUSED_HEAD=0), per slice count."""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mclstexp_amd import ops  # noqa: E402


def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(*s, generator=g) * 2 - 1) * scale).cuda()


def run(M, D, fused, ks, reps=300):
    ops.FUSED_HEAD = fused
    if ks:
        os.environ["MCL_HEAD_KSPLIT"] = str(ks)
    else:
        os.environ.pop("MCL_HEAD_KSPLIT", None)
    q = [rnd(256, D, seed=2, scale=1 / math.sqrt(D)), rnd(256, seed=3, scale=.1), rnd(256, 256, seed=4, scale=1 / 16),
         rnd(256, seed=5, scale=.1), 1 + rnd(256, seed=6, scale=.2), rnd(256, seed=7, scale=.1)]
    for t in q:
        t.requires_grad_(True)
        t.grad = torch.zeros_like(t)          # (the step's parameters own dense .grad buffers: gradients are added in place)
    x = rnd(M, D, seed=1, scale=2.0).requires_grad_(True)
    de = rnd(M, 256, seed=8)

    def step():
        e = ops.ProjectionHeadFn.apply(x, *q)
        e.backward(de)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):          # best of three: single measurements of these 60-100 us graphs come out 2-4 x slow now and then
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="128x1000,128x1024,256x3467,32x1000")
    args = ap.parse_args()
    ops.set_compute("f32")
    for sh in args.shapes.split(","):
        M, D = (int(v) for v in sh.split("x"))
        row = {"M": M, "D": D, "separate_us": round(run(M, D, False, 0), 2)}
        for ks in (0, 2, 4, 8, 16):
            row[f"fused_ks{ks or 'auto'}_us"] = round(run(M, D, True, ks), 2)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

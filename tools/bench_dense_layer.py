#!/usr/bin/env python3
"""Micro-benchmark of the backward launch units of one DenseNet-121 dense layer at the four block shapes of
BASELINE configs[1] (B = 128, 224x224): each C-ABI call alone on the GPU, HIP events, median of repeats.

    python tools/bench_dense_layer.py [--json out.jsonl]

Columns: algorithmic MB (mclstexp_amd/kernel_costs.py), microseconds, achieved TB/s, fraction of the 8 TB/s HBM peak.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mclstexp_amd import _lib, densenet_fused as dn  # noqa: E402

L = _lib.lib()
dev = "cuda"
SHAPES = [(401408, 56, 64, 256), (401408, 56, 224, 256), (100352, 28, 128, 512), (100352, 28, 480, 512),
          (25088, 14, 256, 1024), (25088, 14, 992, 1024), (6272, 7, 512, 1024), (6272, 7, 992, 1024)]


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--only", default="", help="substring filter on the call name")
    args = ap.parse_args()
    rows = []
    g = torch.Generator().manual_seed(0)
    for S, hw, C, ld in SHAPES:
        B = S // (hw * hw)
        xw = ((torch.rand(S, ld, generator=g) - 0.4) * 2).to(torch.bfloat16).to(dev)
        gw = ((torch.rand(S, ld, generator=g) - 0.5) * 0.1).to(torch.bfloat16).to(dev)
        x, gbuf = xw[:, :C], gw[:, :C]
        dz = ((torch.rand(S, 128, generator=g) - 0.5) * 0.2).to(torch.bfloat16).to(dev)
        z = ((torch.rand(S, 128, generator=g) - 0.4) * 2).to(torch.bfloat16).to(dev)
        W1 = ((torch.rand(128, C, generator=g) - 0.5) / 8).to(torch.bfloat16).to(dev)
        W2 = ((torch.rand(32, 3, 3, 128, generator=g) - 0.5) / 8).to(torch.bfloat16).to(dev)
        gam = (torch.rand(1024, generator=g) + 0.5).to(dev)
        bet = (torch.rand(1024, generator=g) - 0.5).to(dev)
        mu = (torch.rand(1024, generator=g) - 0.5).to(dev)
        rs = (torch.rand(1024, generator=g) + 0.5).to(dev)
        dg, db = torch.zeros(1024, device=dev), torch.zeros(1024, device=dev)
        dW1 = torch.zeros(128, C, device=dev)
        dW2 = torch.zeros(32, 3, 3, 128, device=dev)
        coef = torch.zeros(2 * C, device=dev)
        dy = gw[:, ld - 32:]
        scratch, dzo = torch.empty_like(z), torch.empty_like(z)
        ws = torch.empty(max(L.mcl_wrw_workspace_floats(S, 128, C), L.mcl_dense_bn1_bwd_workspace_floats(S, C),
                             L.mcl_dense_conv3x3_wrw_workspace_floats(S), L.mcl_dense_conv3x3_bwd_workspace_floats(S),
                             1 << 20), device=dev)
        st = dn._stream
        P = lambda t: t.data_ptr()
        ym, yv, yr = (torch.zeros(32, device=dev) for _ in range(3))
        ws3 = torch.empty(L.mcl_dense_conv3x3_workspace_floats(S), device=dev)
        zo = torch.empty_like(z)
        zm, zv, zr = (torch.zeros(128, device=dev) for _ in range(3))
        ws1 = torch.empty(L.mcl_dense_conv1x1_workspace_floats(S), device=dev)
        calls = {
            "conv1x1_fwd (+finalize)": (lambda: L.mcl_dense_conv1x1_fwd(P(x), ld, S, C, P(gam), P(bet), P(mu), P(rs), P(W1), P(zo),
                                                                       128, P(ws1), 1e-5, P(zm), P(zv), P(zr), st()),
                                        2 * S * (C + 128)),
            "conv3x3_fwd (+finalize)": (lambda: L.mcl_dense_conv3x3_fwd(P(z), S, hw, hw, P(gam), P(bet), P(mu), P(rs), P(W2),
                                                                       P(xw) + 2 * (ld - 32), ld, P(ws3), 1e-5, P(ym), P(yv),
                                                                       P(yr), st()), 2 * S * 160),
            "bn1_wrw (fused Gram + merge)": (lambda: L.mcl_dense_bn1_wrw(P(dz), P(W1), C, P(x), ld, S, P(gam), P(bet), P(mu), P(rs),
                                                                         P(ws), P(dW1), 1, P(dg), P(db), 1, P(coef), st()),
                                             2 * S * (128 + C)),
            "bn1_dx": (lambda: L.mcl_dense_bn1_dx(P(dz), P(W1), C, P(x), ld, S, P(gam), P(bet), P(mu), P(rs), P(coef), P(gbuf),
                                                  ld, st()), 2 * S * (128 + 3 * C)),
            "r01 bn1_bwd (reduce+fin+dx)": (lambda: L.mcl_dense_bn1_bwd(P(dz), P(W1), C, P(x), ld, S, P(gam), P(bet), P(mu), P(rs),
                                                                        P(ws), P(dg), P(db), 1, P(gbuf), ld, st()),
                                            2 * S * (128 + C) + 2 * S * (128 + 3 * C)),
            "conv1x1_wrw_det (prologue, side)": (lambda: L.mcl_conv1x1_wrw_det(P(dz), 128, P(x), ld, P(gam), P(bet), P(mu), P(rs), P(ws),
                                                                               P(dW1), 1, S, 128, C, st()), 2 * S * (128 + C)),
            "conv3x3_wrw_det": (lambda: L.mcl_dense_conv3x3_wrw_det(P(dy), ld, P(z), S, hw, hw, P(gam), P(bet), P(mu), P(rs), P(ws),
                                                                    P(dW2), 1, st()), 2 * S * 160),
            "conv3x3_bwd (+fin+bn2_dz)": (lambda: L.mcl_dense_conv3x3_bwd(P(dy), ld, S, hw, hw, P(W2), P(z), P(gam), P(bet), P(mu),
                                                                          P(rs), P(ws), P(dg), P(db), 1, P(scratch), P(dzo), st()),
                                          2 * S * (32 + 128 + 128) + 2 * S * 384),
        }
        for name, (fn, nbytes) in calls.items():
            if args.only and args.only not in name:
                continue
            if C != SHAPES[[s[0] for s in SHAPES].index(S)][2] and "3x3" in name:
                continue                                    # the 3x3 kernels do not depend on C_in: once per block
            def run(fn=fn, name=name):
                _lib.check(fn(), name)
            us = timeit(run)
            row = {"S": S, "C_in": C, "call": name, "alg_MB": round(nbytes / 1e6, 1), "us": round(us, 1),
                   "TBps": round(nbytes / us / 1e6, 3), "frac_hbm": round(nbytes / us / 1e6 / 8.0, 3)}
            rows.append(row)
            print(f"S={S:7d} C={C:4d} {name:30s} {row['alg_MB']:8.1f} MB {us:8.1f} us {row['TBps']:6.2f} TB/s "
                  f"({row['frac_hbm']:.2f})", flush=True)
    if args.json:
        with open(args.json, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()

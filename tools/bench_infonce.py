#!/usr/bin/env python3
"""Micro-benchmark of the symmetric InfoNCE kernels (model.py:242-247) at config and scaled shapes.

    python tools/bench_infonce.py [--sizes 1024,2048,...] [--iters 20] [--unfused]

Per size B (P = 256, T = 1): fused bf16 path = cast + 2 x lse + 2 x grad (csrc/infonce_fused.hip), timed with HIP
events on the launch stream.  Reports
  alg TF/s  = 6*B^2*P / t   (logits once + the two gradient contractions: the algorithmic minimum)
  exec TF/s = 12*B^2*P / t  (what the kernels execute: logits recomputed per direction, flash-style)
against the gfx950 dense bf16 MFMA peak (2.5 PF/s).  One JSON line per size on stdout.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="128,1024,2048,4096,8192,16384,32768")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--unfused", action="store_true", help="also time the exact-fp32 unfused path (B <= 8192)")
    a = ap.parse_args()
    from mclstexp_amd import _lib, ops
    _lib.lib()
    P = 256
    for B in [int(s) for s in a.sizes.split(",")]:
        g = torch.Generator(device="cuda").manual_seed(B)
        es = torch.nn.functional.layer_norm(torch.randn(B, P, device="cuda", generator=g), (P,))
        ei = torch.nn.functional.layer_norm(torch.randn(B, P, device="cuda", generator=g) + 0.3 * es, (P,))
        es16, ei16 = ops.cast_bf16(es), ops.cast_bf16(ei)
        rl, _ = ops.infonce_fused_lse(es16, ei16, 1.0)
        cl, _ = ops.infonce_fused_lse(ei16, es16, 1.0)
        coef = 1.0 / (2.0 * B)
        it = max(3, a.iters if B <= 8192 else a.iters // 4)
        t_all = timed(lambda: ops.infonce_fused_fwd_bwd(es, ei, 1.0, min_fused_batch=0), it)
        t_lse = timed(lambda: ops.infonce_fused_lse(es16, ei16, 1.0), it)
        t_grad = timed(lambda: ops.infonce_fused_grad(es16, ei16, 1.0, rl, cl, coef), it)
        fl = 2.0 * B * B * P
        out = {"B": B, "P": P, "fused_ms": round(t_all * 1e3, 4),
               "alg_TFs": round(3 * fl / t_all / 1e12, 2), "exec_TFs": round(6 * fl / t_all / 1e12, 2),
               "frac_of_bf16_peak_alg": round(3 * fl / t_all / 2.5e15, 4),
               "frac_of_bf16_peak_exec": round(6 * fl / t_all / 2.5e15, 4),
               "lse_call_ms": round(t_lse * 1e3, 4), "lse_call_TFs": round(fl / t_lse / 1e12, 2),
               "grad_call_ms": round(t_grad * 1e3, 4), "grad_call_TFs": round(2 * fl / t_grad / 1e12, 2)}
        # fp8 similarity contraction (BASELINE configs[4]): e4m3 LSE kernel (K = 64 MFMA, hardware block scales)
        s8, s16 = ops.quant_e4m3(es)
        i8, i16 = ops.quant_e4m3(ei)
        t8_all = timed(lambda: ops.infonce_fp8_fwd_bwd(es, ei, 1.0), it)
        t8_lse = timed(lambda: ops.infonce_fp8_lse(s8, i8, 1.0), it)
        out.update({"fp8_ms": round(t8_all * 1e3, 4), "fp8_lse_call_ms": round(t8_lse * 1e3, 4),
                    "fp8_lse_call_TFs": round(fl / t8_lse / 1e12, 2),
                    "fp8_lse_frac_of_fp8_peak": round(fl / t8_lse / 5.0e15, 4)})
        if a.unfused and B <= 8192:
            ops.set_compute("f32")
            t_u = timed(lambda: ops.infonce_fwd_bwd(es, ei, 1.0, want_logits=False), max(3, it // 2))
            out["unfused_f32_ms"] = round(t_u * 1e3, 4)
            out["unfused_f32_alg_TFs"] = round(3 * fl / t_u / 1e12, 2)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

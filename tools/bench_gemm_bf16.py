#!/usr/bin/env python3
"""Micro-benchmark of mcl_gemm_bf16 at the ViT-B/16 (batch 256) shapes with the epilogues the encoder block uses (MCL_BENCH_PLAIN_EPILOGUE=1:
plain stores, the round-2..4 records); hipBLASLt column = the bare product."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mclstexp_amd import vit_fused as vf  # noqa: E402

dev, BF = "cuda", torch.bfloat16


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


M = 256 * 197
only = sys.argv[1] if len(sys.argv) > 1 else ""
# (name, M, N, K, flags, epilogue as in the ViT block: bias | bias + GELU + stored pre-activation | bias + residual | gelu')
cases = [("fwd qkv  NT", M, 2304, 768, 0, "bias"), ("fwd fc1  NT", M, 3072, 768, 0, "bias_gelu_pre"),
         ("fwd fc2  NT", M, 768, 3072, 0, "bias_resid"),
         ("dgrad fc1 NN", M, 768, 3072, vf.B_KM, ""), ("dgrad fc2 NN", M, 3072, 768, vf.B_KM, "gelu_bwd"),
         ("wgrad fc1 TT", 3072, 768, M, vf.A_KM | vf.B_KM | vf.OUT_F32, "")]
plain = os.environ.get("MCL_BENCH_PLAIN_EPILOGUE", "0") == "1"
for name, m, n, k, flags, epi in cases:
    if only and only not in name:
        continue
    if plain:
        epi = ""
    akm, bkm = bool(flags & 1), bool(flags & 2)
    A = torch.randn((k, m) if akm else (m, k), device=dev).to(BF)
    B = torch.randn((k, n) if bkm else (n, k), device=dev).to(BF)
    f32 = bool(flags & 16)
    C = torch.empty((m, n), device=dev, dtype=torch.float32 if f32 else BF)
    ks = vf._ksplit(m, n) if f32 else 1
    kw = {}
    if "bias" in epi:
        kw["bias"] = torch.randn(n, device=dev)
    if "gelu_pre" in epi:
        flags |= vf.GELU | (0 if os.environ.get("MCL_BENCH_OLD_GELU") == "1" else vf.GELU_GRAD_OUT)
        kw["pre_out"] = torch.empty((m, n), device=dev, dtype=BF)
        kw["ldp"] = n
    if "resid" in epi:
        kw["resid"] = torch.randn((m, n), device=dev).to(BF)
        kw["ldr"] = n
    if epi == "gelu_bwd":
        flags |= vf.GELU_BWD if os.environ.get("MCL_BENCH_OLD_GELU") == "1" else vf.AUX_IS_GRAD
        kw["aux"] = torch.randn((m, n), device=dev).to(BF)
        kw["ldaux"] = n
    t = timeit(lambda: vf.gemm(A, B, C, m, n, k, A.shape[1], B.shape[1], n, flags=flags, ksplit=ks, accumulate=False, **kw))
    tt = timeit(lambda: torch.mm(A.t() if akm else A, B if bkm else B.t())) if not f32 else float("nan")
    print(json.dumps({"case": name, "epilogue": epi or "plain", "M": m, "N": n, "K": k, "ms": round(t * 1e3, 3),
                      "TFs": round(2.0 * m * n * k / t / 1e12, 1),
                      "hipblaslt_ms": None if f32 else round(tt * 1e3, 3),
                      "hipblaslt_TFs": None if f32 else round(2.0 * m * n * k / tt / 1e12, 1)}), flush=True)

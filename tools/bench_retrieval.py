#!/usr/bin/env python3
"""Micro-benchmark of the inference-time retrieval path on one MI355X: per-stage times (HIP events),
the similarity GEMM's fp32-MFMA rate, the top-k kernel's effective row-read bandwidth, torch.topk on the same
matrix for scale, and (bounded) the numpy oracle on the host.  One JSON line per shape.

    python tools/bench_retrieval.py [--cpu]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mclstexp_amd import retrieval as rt, synth  # noqa: E402

SHAPES = [  # name, N keys, Q queries, genes, top_k, ord
    ("her2st fold (31 slides of keys)", 13000, 450, 785, 200, 1),
    ("cscc fold", 8000, 700, 171, 600, 2),
    ("visium fold", 9000, 3600, 1000, 200, 2),
    ("scaled", 100000, 4096, 1000, 600, 2),
]


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true", help="also time the numpy oracle on a bounded number of queries")
    args = ap.parse_args()
    dev = "cuda"
    for name, n, q, g, k, ord_ in SHAPES:
        c = synth.make_retrieval_case(n, q, 256, g, seed=11)
        key, qry, expr = (torch.from_numpy(c[x]).to(dev) for x in ("spot_key", "image_query", "expression_key"))
        kn, qn = rt.l2_normalize(key), rt.l2_normalize(qry)
        sim = rt.cosine_similarity_matrix(qn, kn)
        _, idx = rt.topk_rows(sim, k)
        t_norm = timeit(lambda: (rt.l2_normalize(key), rt.l2_normalize(qry)))
        t_gemm = timeit(lambda: rt.cosine_similarity_matrix(qn, kn))
        t_topk = timeit(lambda: rt.topk_rows(sim, k))
        t_avg = timeit(lambda: rt.weighted_average_device(key, expr, qry, idx, ord_))
        t_all = timeit(lambda: rt.weighted_average_device(key, expr, qry, rt.find_matches_device(key, qry, k)[1], ord_))
        t_torch = timeit(lambda: torch.topk(sim, k), iters=5)
        # similarity + top-k without the (Q, N) matrix in HBM (retrieval.find_matches_filtered) against GEMM + top-k on it
        t_fused = redo = None
        if n >= rt.FUSED_MIN_KEYS and 16 * k <= n:
            redo = rt.find_matches_filtered(qn, kn, k)[2]
            t_fused = timeit(lambda: rt.find_matches_filtered(qn, kn, k))
        rec = {"shape": name, "N": n, "Q": q, "G": g, "top_k": k, "ord": ord_,
               "normalize_ms": round(t_norm, 4), "similarity_gemm_ms": round(t_gemm, 4),
               "similarity_TFs_fp32": round(2.0 * q * n * 256 / t_gemm / 1e9, 1),
               "topk_ms": round(t_topk, 4), "topk_row_GBs": round(4.0 * q * n / t_topk / 1e6, 1),
               "torch_topk_ms": round(t_torch, 4),
               "similarity_plus_topk_materialised_ms": round(t_gemm + t_topk, 4),
               "similarity_plus_topk_filtered_ms": None if t_fused is None else round(t_fused, 4),
               "filtered_rows_recomputed": redo,
               "weighted_average_ms": round(t_avg, 4),
               "gather_GBs": round(4.0 * q * k * (g + 2 * 256) / t_avg / 1e6, 1),
               "end_to_end_ms": round(t_all, 4), "queries_per_s": round(q / t_all * 1e3, 0)}
        if args.cpu:
            from oracle import ref_retrieval
            qs = min(q, 64)
            t0 = time.perf_counter()
            _, ii = ref_retrieval.find_matches(c["spot_key"], c["image_query"][:qs], top_k=k)
            ref_retrieval.weighted_prediction(c["spot_key"], c["expression_key"], c["image_query"][:qs], ii, ord=ord_)
            dt = time.perf_counter() - t0
            rec["cpu_oracle_queries_per_s"] = round(qs / dt, 1)
            rec["cpu_sample"] = f"{qs} queries, torch {torch.get_num_threads()} threads + numpy loop"
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()

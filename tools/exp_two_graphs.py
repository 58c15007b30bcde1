#!/usr/bin/env python3
"""Two single-stream HIP graphs replayed at the same time on two streams: do they run concurrently, and at the
single-stream (batched) per-kernel cost?  Compare with tools/exp_graph_branches.py (one graph with two branches)."""
import json
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n_elem = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
dev = "cuda"


def chain(x, n):
    for _ in range(n):
        x = torch.sin(x)
    return x


a = torch.rand(n_elem, device=dev)
b = torch.rand(n_elem, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=s1):
    ya = chain(a, N)
with torch.cuda.graph(g2, stream=s2):
    yb = chain(b, N)


def run_both():
    with torch.cuda.stream(s1):
        g1.replay()
    with torch.cuda.stream(s2):
        g2.replay()


def run_one():
    with torch.cuda.stream(s1):
        g1.replay()


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(json.dumps({"N": N, "elements": n_elem, "one_graph_ms": round(timed(run_one), 4),
                  "two_graphs_two_streams_ms": round(timed(run_both), 4)}))

#!/usr/bin/env python3
"""Do two independent branches of a captured HIP graph really run concurrently on this ROCm?  Two chains of N small
kernels (each a handful of workgroups, ~5-10 us, dependent within the chain) are captured (a) on one stream, (b) on two
streams forked from the capture stream and joined at the end, (c) like (b) but with a cross-stream event every K kernels
(the pattern of the weight-gradient side stream).  Prints replay time per variant."""
import json
import sys
import time

import torch

dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_elem = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16


def chain(x, n):
    for _ in range(n):
        x = torch.sin(x)          # one small kernel per step, dependent on the previous one
    return x


def timed(g, reps=50):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


a = torch.rand(n_elem, device=dev)
b = torch.rand(n_elem, device=dev)
side = torch.cuda.Stream()
res = {}

g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    ya = chain(a, N)
res["one_branch_N"] = timed(g)

g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    ya = chain(a, N)
    yb = chain(b, N)
res["serial_2N"] = timed(g)

g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        yb = chain(b, N)
    ya = chain(a, N)
    main.wait_stream(side)
res["two_branches"] = timed(g)

for K in (10, 1):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        xa, xb = a, b
        for i in range(N // K):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                xb = chain(xb, K)
            xa = chain(xa, K)
            main.wait_stream(side)
    res[f"fork_join_every_{K}"] = timed(g)

# unequal branches: long main chain (N), short side chain (N/4) forked at the start, joined at the end
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        yb = chain(b, N // 4)
    ya = chain(a, N)
    main.wait_stream(side)
res["long_main_short_side"] = timed(g)
# a long single-stream chain followed by ONE small forked section: is the branch overhead local or global?
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    main = torch.cuda.current_stream()
    ya = chain(a, N)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        yb = chain(b, 5)
    ya = chain(ya, 5)
    main.wait_stream(side)
res["chain_N_then_small_fork"] = timed(g)

# two graphs: single-stream chain (N) in one, the forked pair in another, replayed back to back
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    ya = chain(a, N)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        yb = chain(b, 5)
    yc = chain(a, 5)
    main.wait_stream(side)


class Two:
    def replay(self):
        g1.replay()
        g2.replay()


res["two_graphs_chain_then_fork"] = timed(Two())
print(json.dumps({"N": N, "elements": n_elem, "ms": {k: round(v, 4) for k, v in res.items()}}))

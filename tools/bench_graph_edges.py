#!/usr/bin/env python3
"""Cost of cross-stream edges inside a captured HIP graph (what engine/densenet_fused scheduling is built around).
A chain of N small dependent kernels on the capture stream; variants:
  chain      no side stream
  fork       every kernel also forks one side kernel (event record + side wait), joined once at the end
  join_lag   as fork, and the main chain waits, before kernel i, for the side kernel forked at i - LAG (long finished)
  join_now   as fork, and the main chain waits for the side kernel forked at i (fork + immediate join)
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N, LAG = 200, 4
x = torch.zeros(1 << 16, device="cuda")
ys = [torch.zeros(1 << 14, device="cuda") for _ in range(N)]
side = torch.cuda.Stream()


def build(mode):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        evs = []
        for i in range(N):
            if mode == "join_lag" and i >= LAG:
                main.wait_event(evs[i - LAG])
            x.add_(1.0)                                   # the dependent main chain
            if mode != "chain":
                e = torch.cuda.Event(); e.record(main)
                side.wait_event(e)
                with torch.cuda.stream(side):
                    ys[i].add_(1.0)
                    d = torch.cuda.Event(); d.record(side); evs.append(d)
                if mode == "join_now":
                    main.wait_event(d)
        if mode != "chain":
            main.wait_stream(side)
    return g


for mode in ("chain", "fork", "join_lag", "join_now"):
    g = build(mode)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{mode:9s} {e0.elapsed_time(e1) / 10 * 1e3 / N:7.2f} us per chain kernel", flush=True)

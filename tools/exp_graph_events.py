#!/usr/bin/env python3
"""Experiment: can timing events recorded INSIDE a captured HIP graph be read after replay?
(torch.cuda.Event(external=True) -> hipEventRecordWithFlags(hipEventRecordExternal))."""
import torch

dev = torch.device("cuda")
x = torch.randn(8192, 8192, device=dev)
y = torch.empty_like(x)
s = torch.cuda.Stream()
for external in (True, False):
    try:
        evs = [torch.cuda.Event(enable_timing=True, external=external) for _ in range(4)]
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            evs[0].record()
            torch.mul(x, 2.0, out=y)
            evs[1].record()
            evs[2].record()
            torch.add(x, 1.0, out=y)
            evs[3].record()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print(f"external={external}: mul {evs[0].elapsed_time(evs[1]):.4f} ms, mm {evs[2].elapsed_time(evs[3]):.4f} ms")
    except Exception as e:  # noqa
        print(f"external={external}: FAILED {type(e).__name__}: {str(e)[:300]}")
# eager reference
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.mul(x, 2.0, out=y); e1.record(); torch.cuda.synchronize()
print(f"eager mul {e0.elapsed_time(e1):.4f} ms")
e0.record(); torch.add(x, 1.0, out=y); e1.record(); torch.cuda.synchronize()
print(f"eager mm {e0.elapsed_time(e1):.4f} ms")

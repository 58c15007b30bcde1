"""Which GPU kernels does a training step launch?  (the accounting must come from the kernel list the GPU
actually ran, not from ``densenet_fused.fallback_counts()``.)

``step_kernels(fn)`` runs ``fn`` under ``torch.profiler`` (kernel activity records: eager launches AND the kernels of a
replayed HIP graph) and returns {kernel name: launches}.  ``foreign(names)`` filters the names that are not this
library's: hipBLASLt / Tensile (``Cijk_*``), ATen (``at::native::*``, ``at::cuda::*``), MIOpen, rocPRIM / hipCUB.
Used by tests/test_own_kernels_gpu.py, bench.py (``config.foreign_kernels``) and tools/list_step_kernels.py.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List

import torch

FOREIGN_MARKERS = ("Cijk_", "at::native", "at::cuda", "at_cuda", "miopen", "MIOpen", "rocprim", "hipcub", "void at::",
                   "Tensile", "rocblas", "hipblas")
# copy engine / runtime helpers that are not compute kernels of any library (graph memcpy / memset nodes)
RUNTIME_HELPERS = ("__amd_rocclr_copyBuffer", "__amd_rocclr_fillBuffer", "Memcpy", "Memset")


def step_kernels(fn: Callable[[], None]) -> Dict[str, int]:
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    out: Dict[str, int] = {}
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            out[ev.name] = out.get(ev.name, 0) + 1
    return out


def foreign(names: Iterable[str]) -> List[str]:
    bad = []
    for n in names:
        if any(h in n for h in RUNTIME_HELPERS):
            continue
        if any(m in n for m in FOREIGN_MARKERS):
            bad.append(n)
    return sorted(bad)

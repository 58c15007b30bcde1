"""Micro-benchmark: 1x1-conv weight gradient, csrc/conv1x1.hip vs MIOpen (aten.convolution_backward)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mclstexp_amd import _lib, densenet_fused as dn
L = _lib.lib()
dev = "cuda"
shapes = [(401408, 64), (401408, 224), (100352, 128), (100352, 480), (25088, 256), (25088, 992), (6272, 512), (6272, 992)]
HW = {401408: 56, 100352: 28, 25088: 14, 6272: 7}


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'S':>7} {'Cin':>5} {'MB':>7} {'hip us':>8} {'TB/s':>6} {'miopen us':>10}")
for S, C in shapes:
    hw = HW[S]; B = S // (hw * hw)
    dz = (torch.rand(B, 128, hw, hw, device=dev) - 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    a = (torch.rand(B, C, hw, hw, device=dev) - 0.3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(128, C, 1, 1, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dW = torch.zeros(128, C, device=dev)
    f1 = lambda: _lib.check(L.mcl_conv1x1_wrw_bf16(dz.data_ptr(), 128, a.data_ptr(), C, None, None, None, None, dW.data_ptr(), C, S, 128, C, dn._stream()))
    f2 = lambda: torch.ops.aten.convolution_backward(dz, a, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])
    t1, t2 = timeit(f1), timeit(f2)
    mb = S * (128 + C) * 2 / 1e6
    print(f"{S:7d} {C:5d} {mb:7.1f} {t1:8.1f} {mb / t1:6.2f} {t2:10.1f}")

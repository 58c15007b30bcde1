#!/usr/bin/env python3
"""UNTRACED timeline of the replayed step graph: GPU wall-clock stamps (mcl_stamp, one-thread kernels captured into the graph at
labelled points of both lanes) instead of a kernel trace -- the tracer inflates launches and changes how the graph's two
hardware queues interleave.      MCL_STAMPS=1 python tools/step_stamps.py [--steps 30]"""
import argparse, os, sys
os.environ["MCL_STAMPS"] = "1"
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mclstexp_amd import densenet_fused as dn, synth
from mclstexp_amd.engine import TrainStep
from mclstexp_amd.model import mclSTExp_Attention
from mclstexp_amd.optim import FusedAdam

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=30); a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = mclSTExp_Attention("densenet121", 1.0, 1024, 1000, 256, 8, 64, 2, backbone_dtype=torch.bfloat16, embedding_grad="rowsparse",
                       infonce="fused").to(dev)
m.to(memory_format=torch.channels_last).train()
opt = FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3).attach_model(m)
b = {k: v.to(dev) for k, v in synth.make_batch(128, 1000, image_hw=224, seed=0).items()}
b["image"] = b["image"].contiguous(memory_format=torch.channels_last)
tr = TrainStep(m, opt, None, graphs=True, warmup=3)
for _ in range(8):
    tr(b)
acc = {}
for _ in range(a.steps):
    tr(b)
    st = dn.read_stamps()
    t0 = st["step start (main)"]
    for k, v in st.items():
        acc.setdefault(k, []).append(v - t0)
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(20):
    tr(b)
torch.cuda.synchronize()
print(f"step (with {len(acc)} stamp launches): {(time.perf_counter() - t) / 20 * 1e3:.3f} ms")
for k, v in sorted(acc.items(), key=lambda kv: sum(kv[1]) / len(kv[1])):
    v = sorted(v)
    print(f"{v[len(v) // 2]:10.1f} us   {k}")

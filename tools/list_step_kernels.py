#!/usr/bin/env python3
"""Kernel names of ONE training step of the benched configuration (eager and graph replay), from torch.profiler's kernel
activity records; library kernels (hipBLASLt / ATen / MIOpen) are marked with '!'.

    python tools/list_step_kernels.py [--batch 128 --image 224 --genes 1000]
"""
import argparse
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--image", type=int, default=224)
    ap.add_argument("--genes", type=int, default=1000)
    a = ap.parse_args()
    from mclstexp_amd import kernel_audit, synth
    from mclstexp_amd.engine import TrainStep
    from mclstexp_amd.model import mclSTExp_Attention
    from mclstexp_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = mclSTExp_Attention("densenet121", 1.0, 1024, a.genes, 256, 8, 64, 2, backbone_dtype=torch.bfloat16,
                               embedding_grad="rowsparse", infonce="fused").to(dev)
    model.to(memory_format=torch.channels_last).train()
    opt = FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-3).attach_model(model)
    b = {k: v.to(dev) for k, v in synth.make_batch(a.batch, a.genes, image_hw=a.image, seed=0).items()}
    b["image"] = b["image"].contiguous(memory_format=torch.channels_last)
    tr = TrainStep(model, opt, None, graphs=False)
    for mode, fn in (("reference loop (autograd root)", lambda: tr(b)), ("captured sequence, eager", lambda: tr.run_sequence_eager(b))):
        for _ in range(4):
            fn()
        ks = kernel_audit.step_kernels(fn)
        bad = set(kernel_audit.foreign(ks))
        print(f"== {mode}: {sum(ks.values())} kernel records, {len(ks)} distinct, {sum(ks[n] for n in bad)} foreign launches")
        for n, c in sorted(ks.items(), key=lambda kv: (kv[0] not in bad, -kv[1])):
            print(f"{'!' if n in bad else ' '} {c:5d}  {n[:200]}")


if __name__ == "__main__":
    main()

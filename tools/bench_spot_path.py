#!/usr/bin/env python3
"""Serial time of the spot branch: one training step of mclSTExp_Attention with the identity image encoder (features
supplied), i.e. spot encoder + both projection heads + InfoNCE + Adam incl. the position tables, replayed as ONE HIP
graph on one stream.  A/B the split-K path of mcl_gemm with MCL_GEMM_SPLITK=0."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mclstexp_amd import ops, synth  # noqa: E402
from mclstexp_amd.engine import TrainStep  # noqa: E402
from mclstexp_amd.model import mclSTExp_Attention  # noqa: E402
from mclstexp_amd.optim import FusedAdam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--genes", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--no_tables", action="store_true", help="stock dense embedding gradients off: rowsparse anyway")
    args = ap.parse_args()
    G, D, B = args.genes, 1024, args.batch
    m = mclSTExp_Attention("identity", 1.0, D, G, 256, 8, 64, 2, embedding_grad="rowsparse", infonce="fused")
    m.load_state_dict(synth.make_params(G, D, seed=0))
    m.cuda().train()
    opt = FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3).attach_model(m)
    step = TrainStep(m, opt, None, graphs=True, warmup=2)
    batches = [{k: v.cuda() for k, v in synth.make_batch(B, G, image_dim=D, seed=s).items()} for s in range(4)]
    for i in range(10):
        step(batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(batches[i % 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    from mclstexp_amd import kernel_audit
    ks = kernel_audit.step_kernels(lambda: step(batches[0]))
    ks = {k: v for k, v in ks.items() if not any(h in k for h in kernel_audit.RUNTIME_HELPERS)}
    print(json.dumps({"workload": "spot branch + heads + InfoNCE + Adam (identity image encoder), one stream",
                      "batch": B, "genes": G, "split_k": ops.SPLIT_K, "fused_head": ops.FUSED_HEAD,
                      "ms_per_step": round(dt * 1e3, 4), "kernel_launches_per_step": sum(ks.values()),
                      "kernels": dict(sorted(((k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:], v)
                                               for k, v in ks.items()), key=lambda kv: -kv[1]))}),
          flush=True)


if __name__ == "__main__":
    main()

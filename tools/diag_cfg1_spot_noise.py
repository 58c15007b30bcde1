#!/usr/bin/env python3
"""Where does the step-2..4 spot-embedding deviation of tests/test_configs_gpu.py::test_cfg1_as_benched_vs_oracle come from?
(the bound was widened from 0.1 to 0.06*s next to three kernel changes.)

Runs the CPU oracle once (4 steps, configs[1]) and then the GPU step in several variants against it:
  benched            bf16 backbone kernels, fused InfoNCE, step graph (what the test asserts)
  two_pass_bn1       the same with the single-pass BatchNorm-1 backward off (dn.USE_BN1_SINGLE_PASS = False)
  fp32_backbone      fp32 activations (library convolutions), exact InfoNCE, eager: NO bf16 kernel anywhere -- what is left is
                     Adam's +-lr sign noise between two fp32 executions with different summation orders
Prints max|dE_spot| per step and variant.
"""
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402


def main():
    import test_configs_gpu as T
    from mclstexp_amd import densenet_fused as dn, synth
    B, G, HW, steps = 128, 1000, 224, 4
    batches = [synth.make_batch(B, G, image_hw=HW, seed=s) for s in range(steps)]
    torch.set_num_threads(min(32, torch.get_num_threads()))
    _, params = T._model_and_oracle_params(G, backbone_dtype=torch.bfloat16, infonce="fused")
    state, ref = {}, []
    t0 = time.time()
    for s, b in enumerate(batches):
        ref.append(T._oracle_step(params, state, b, s + 1))
    print(f"oracle: {(time.time() - t0) / steps:.1f} s/step", flush=True)
    variants = [("benched", dict(backbone_dtype=torch.bfloat16, infonce="fused"), True, {}),
                ("two_pass_bn1", dict(backbone_dtype=torch.bfloat16, infonce="fused"), True, {"USE_BN1_SINGLE_PASS": False}),
                ("fp32_backbone", dict(backbone_dtype=None, infonce="exact"), False, {})]
    for name, kw, graphs, attrs in variants:
        saved = {k: getattr(dn, k) for k in attrs}
        for k, v in attrs.items():
            setattr(dn, k, v)
        try:
            m, _ = T._model_and_oracle_params(G, **kw)
            outs, _, _ = T._run_steps(m, batches, graphs=graphs, warmup=2)
        finally:
            for k, v in saved.items():
                setattr(dn, k, v)
        row = []
        for s in range(steps):
            de_s = float((outs[s]["spot_embeddings"] - ref[s]["spot_embeddings"]).abs().max())
            d_i = outs[s]["image_embeddings"] - ref[s]["image_embeddings"]
            row.append(f"step {s + 1}: dE_spot {de_s:.3e} dE_img rms {float(d_i.pow(2).mean().sqrt()):.3e} "
                       f"loss_rel {abs(outs[s]['loss'] - float(ref[s]['loss'])) / max(1.0, abs(float(ref[s]['loss']))):.2e}")
        print(f"{name:14s} " + " | ".join(row), flush=True)
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

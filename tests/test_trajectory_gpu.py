"""SURVEY section 4 "Integration": an N-step training trajectory of the BENCHED mode (bf16 own backbone kernels, one HIP graph incl.
FusedAdam) beside the fp32 own-kernel mode (the reference's numerics: fp32 activations, exact-fp32 MFMA convolutions, exact InfoNCE)
-- same initial weights, the same 16 batches of 128 (patch, spot) pairs cycled for 200 steps (BASELINE configs[1]: 224 x 224,
G = 1000, DenseNet-121; /root/reference/train.py:30-42 with Adam(1e-4, weight_decay 1e-3), train.py:118-120).

The 4-step oracle comparison (test_configs_gpu.py::test_cfg1_as_benched_vs_oracle) shows the two precisions separating step by
step (Adam's first updates are +-lr sign(g)); what matters for a training run is that they CONVERGE alike: the smoothed loss curves
and the in-batch retrieval top-1 accuracy (the model's own use, evel_her2st.py:74-84) after every pass over the batches.
The curves of the run are written to gpurun_out/trajectory.json (kept as profiles/r06_trajectory.json)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, G, HW, STEPS, NB = 128, 1000, 224, 200, 16
CHECK = (16, 32, 48, 64, 100, 200)


def _run(batches, backbone_dtype, infonce, graphs):
    from mclstexp_amd import densenet_fused as dn, synth
    from mclstexp_amd.engine import TrainStep
    from mclstexp_amd.model import mclSTExp_Attention
    from mclstexp_amd.optim import FusedAdam
    torch.manual_seed(0)
    m = mclSTExp_Attention("densenet121", 1.0, 1024, G, 256, 8, 64, 2, embedding_grad="rowsparse",
                           backbone_dtype=backbone_dtype, infonce=infonce)
    sd = m.state_dict()
    sd.update(synth.make_params(G, 1024, seed=0))
    m.load_state_dict(sd)
    m.to(DEV)
    if backbone_dtype is not None:
        m.to(memory_format=torch.channels_last)
    m.train()
    m.capture = True
    opt = FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3).attach_model(m)
    tr = TrainStep(m, opt, None, graphs=graphs, warmup=2)
    losses, acc = [], []
    for s in range(STEPS):
        b = batches[s % NB]
        if backbone_dtype is not None and not b["image"].is_contiguous(memory_format=torch.channels_last):
            b["image"] = b["image"].contiguous(memory_format=torch.channels_last)
        losses.append(tr(b))
        es, ei = m.last["spot_embeddings"].float(), m.last["image_embeddings"].float()
        acc.append(((es @ ei.t()).argmax(dim=1) == torch.arange(B, device=DEV)).float().mean())
    tr.check_errors()
    losses = [float(x) for x in torch.stack(losses).cpu()]
    acc = [float(x) for x in torch.stack(acc).cpu()]
    dn.set_weight_provider(None)
    return losses, acc


def _smooth(x, at, w=NB):
    return sum(x[at - w:at]) / w          # one pass over the 16 batches ending at step `at`


def test_bf16_and_fp32_modes_converge_alike_over_200_steps():
    from mclstexp_amd import synth
    batches = [{k: v.to(DEV) for k, v in synth.make_batch(B, G, image_hw=HW, seed=s).items()} for s in range(NB)]
    l16, a16 = _run([dict(b) for b in batches], torch.bfloat16, "fused", graphs=True)      # the mode bench.py times
    l32, a32 = _run([dict(b) for b in batches], None, "exact", graphs=False)               # the reference's numerics
    rec = {"config": f"B={B} G={G} {HW}x{HW} densenet121, {NB} fixed batches cycled, Adam(1e-4, wd 1e-3), {STEPS} steps",
           "loss_bf16_benched_mode": [round(v, 5) for v in l16], "loss_fp32_own_kernels": [round(v, 5) for v in l32],
           "inbatch_top1_bf16": [round(v, 4) for v in a16], "inbatch_top1_fp32": [round(v, 4) for v in a32], "checkpoints": {}}
    for at in CHECK:
        s16, s32 = _smooth(l16, at), _smooth(l32, at)
        t16, t32 = _smooth(a16, at), _smooth(a32, at)
        rec["checkpoints"][str(at)] = {"loss_bf16": round(s16, 5), "loss_fp32": round(s32, 5), "top1_bf16": round(t16, 4),
                                       "top1_fp32": round(t32, 4)}
        print(f"step {at}: smoothed loss bf16 {s16:.4f} fp32 {s32:.4f} (rel {abs(s16 - s32) / s32:.3f}); "
              f"in-batch top-1 bf16 {t16:.3f} fp32 {t32:.3f}")
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "trajectory.json"), "w") as f:
            json.dump(rec, f)
    except OSError:
        pass
    # both modes learn the 2048 pairs (the loss of a pass over the 16 batches falls from ~34 to < 5e-3, retrieval reaches 100 %) ...
    c = rec["checkpoints"]
    assert c["16"]["loss_bf16"] > 20 and c["16"]["loss_fp32"] > 20
    for at in ("100", "200"):
        assert c[at]["loss_bf16"] < 5e-3 and c[at]["loss_fp32"] < 5e-3, c[at]
        assert c[at]["top1_bf16"] >= 0.995 and c[at]["top1_fp32"] >= 0.995, c[at]
    # ... and alike on the way there (measured, one run: pass 1 34.36 / 34.47; pass 2 2.80 / 2.18 with top-1 0.61 / 0.68; pass 3
    # 0.375 / 0.362 with 0.899 / 0.907; pass 4 0.077 / 0.046 with 0.972 / 0.982): the first pass within 2 %, while the loss falls by
    # an order of magnitude per pass the bf16 run trails the fp32 one by less than half a pass -- loss within a factor 2, accuracy
    # within 0.1
    assert abs(c["16"]["loss_bf16"] - c["16"]["loss_fp32"]) <= 0.02 * c["16"]["loss_fp32"], c["16"]
    for at in ("32", "48"):
        assert 0.5 <= c[at]["loss_bf16"] / c[at]["loss_fp32"] <= 2.0, (at, c[at])
        assert abs(c[at]["top1_bf16"] - c[at]["top1_fp32"]) <= 0.10, (at, c[at])
    # pass 4: the losses are a few hundredths (0.077 / 0.037 ... 0.046 in two runs: the fp32 mode's split-K order is not fixed run to
    # run, and at this level single pairs dominate the mean) -- bounded absolutely, the retrieval accuracy within 0.05
    assert c["64"]["loss_bf16"] < 0.25 and c["64"]["loss_fp32"] < 0.25, c["64"]
    assert abs(c["64"]["top1_bf16"] - c["64"]["top1_fp32"]) <= 0.05, c["64"]

"""BASELINE.json configs at their FULL sizes on the MI355X.

  configs[0]  her2st: batch 8, 112x112 patches, DenseNet-121            -> test_cfg0_*
  configs[1]  batch 128, 224x224, 1000 genes, DenseNet-121, bf16, AS BENCHED (bf16 backbone kernels + fused bf16
              InfoNCE + HIP-graph replay + FusedAdam)                      -> test_cfg1_as_benched_vs_oracle
  configs[2]  ViT-B/16 (and the reference's default B/32), batch 256        -> test_cfg2_vit_*
  configs[4]  per-GPU shape: batch 256, 3467 genes, 256x256 patches         -> fixture b256_g3467 (test_model_gpu /
              test_oracle_golden, from the reference's own classes) + test_cfg4_backbone_256px_*

Oracle = oracle/ref_cpu.py (pinned to the reference by tests/test_oracle_golden.py), fp32 on the host cores.
Tolerances of the bf16 modes are stated where they are asserted; they bound the deviation of exactly the
configuration bench.py times.  The DenseNet restatement itself is parity-unpinned (torchvision absent).
"""
import copy
import time

import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model_and_oracle_params(G, encoder="densenet121", **kw):
    from mclstexp_amd import synth
    from mclstexp_amd.model import mclSTExp_Attention
    torch.manual_seed(0)
    m = mclSTExp_Attention(encoder, 1.0, 1024, G, 256, 8, 64, 2, embedding_grad="rowsparse", **kw)
    sd = m.state_dict()
    sd.update(synth.make_params(G, 1024, seed=0))
    m.load_state_dict(sd)
    params = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()
              if v.dtype == torch.float32 and "running_" not in k}
    return m, params


def _oracle_step(params, state, batch, step):
    """ref_cpu.train_step, keeping the embeddings (same statements: forward, autograd backward, Adam over all)."""
    from oracle import ref_cpu
    for p in params.values():
        p.grad = None
    feats = ref_cpu.densenet121_features(params, batch["image"])
    out = ref_cpu.forward_from_features(params, feats, batch["expression"], batch["position"], 1.0, 2, 8, 64)
    out["loss"].backward()
    with torch.no_grad():
        for n, p in params.items():
            if p.grad is None:
                continue
            if n not in state:
                state[n] = (torch.zeros_like(p), torch.zeros_like(p))
            ref_cpu.adam_l2_step(p, p.grad, state[n][0], state[n][1], step)
    return {k: out[k].detach().clone() for k in ("loss", "spot_embeddings", "image_embeddings", "cos_smi")}


def _run_steps(m, batches, graphs, warmup=2):
    from mclstexp_amd import densenet_fused as dn
    from mclstexp_amd.engine import TrainStep
    from mclstexp_amd.optim import FusedAdam
    m.to(DEV)
    if m.backbone_dtype is not None:
        m.to(memory_format=torch.channels_last)
    m.train()
    m.capture = True
    dn.reset_fallbacks()
    opt = FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3).attach_model(m)
    tr = TrainStep(m, opt, None, graphs=graphs, warmup=warmup)
    outs = []
    for i, b in enumerate(batches):
        if i == 1:
            # the FIRST step runs before FusedAdam has built its flat gradient bucket (no .grad to accumulate into yet):
            # its weight gradients take the library hand-over once.  From step 2 on nothing may leave the HIP kernels.
            dn.reset_fallbacks()
        bd = {k: v.to(DEV) for k, v in b.items()}
        if m.backbone_dtype is not None:
            bd["image"] = bd["image"].contiguous(memory_format=torch.channels_last)
        loss = tr(bd)
        outs.append({"loss": float(loss.item()), "spot_embeddings": m.last["spot_embeddings"].float().cpu().clone(),
                     "image_embeddings": m.last["image_embeddings"].float().cpu().clone()})
    fb = dn.fallback_counts()
    dn.set_weight_provider(None)
    return outs, tr, fb


def _worst_param_diff(m, params):
    worst, name = 0.0, ""
    for n, p in m.named_parameters():
        if n.startswith("image_encoder") or "embed" in n:
            continue
        d = float((p.detach().cpu() - params[n].detach()).abs().max())
        if d > worst:
            worst, name = d, n
    return worst, name


# ------------------------------------------------------------------------------------------------ configs[1]
def test_cfg1_as_benched_vs_oracle():
    """B = 128, 224x224, G = 1000, DenseNet-121: bf16 backbone kernels + fused bf16 InfoNCE + HIP-graph replay +
    FusedAdam -- the exact mode bench.py times -- against the fp32 CPU oracle fed the same weights and batches, 4
    steps (2 eager warm-up calls of engine.TrainStep, the capture, one more replay).

    Stated bf16 tolerances (measured deviations are printed):
      * embeddings (LayerNorm-ed rows of norm 16, elements O(1)): image side rms 0.25 / max 1.5 absolute over the
        128 x 256 elements (measured: rms 0.14, max 0.57 -- a random-init 121-layer BatchNorm net amplifies bf16
        rounding chaotically; the stock bf16-autocast module path deviates from fp64 just as much, which is what
        test_cfg4_backbone_256px_accuracy_vs_fp64 and test_fused_densenet_as_accurate_as_module_path pin); spot side 2e-3
        at step 1 (fp32 kernels; measured 1.8e-6).  Later steps inherit the bf16 noise of the IMAGE embeddings through the
        loss gradient: Adam's first updates are +-lr * sign(g), so a spot-path weight whose tiny gradient changes sign moves
        the other way, and a 1000-term fan-in of such weights shows up in the spot embeddings.  Measured (round 4,
        tools/diag_cfg1_spot_noise.py, one box, same seeds) 0.054 / 0.070 / 0.109 at steps 2 / 3 / 4; with the two-pass
        BatchNorm-1 backward instead of the single-pass one 0.054 / 0.071 / 0.103 (the r03 kernel change is NOT the source:
        ADVICE r03), and with an fp32 backbone (no bf16 kernel anywhere, exact InfoNCE) 2.5e-5 / 9.4e-4 / 1.5e-3 -- the
        deviation is the bf16 backbone's 0.14-rms image-embedding noise amplified by Adam, not a spot-path kernel error.
        Bounds = 1.3 x the measured values per step;
      * loss: 3 % of max(1, |loss|) -- logits reach +-85 and the bf16 image embeddings move them by ~0.5;
      * non-backbone parameters after 4 Adam steps: 8.5e-4 absolute = 4 steps x 2 lr (Adam's first updates are
        ~ +-lr * sign(g): an element whose tiny gradient flips sign under bf16 noise moves the other way, so two
        trajectories separate by up to 2 lr per step; measured 7.9e-4) + fp32 noise.
    """
    from mclstexp_amd import synth
    B, G, HW, steps = 128, 1000, 224, 4
    m, params = _model_and_oracle_params(G, backbone_dtype=torch.bfloat16, infonce="fused")
    batches = [synth.make_batch(B, G, image_hw=HW, seed=s) for s in range(steps)]
    torch.set_num_threads(min(32, torch.get_num_threads()))
    state, ref = {}, []
    t0 = time.time()
    for s, b in enumerate(batches):
        ref.append(_oracle_step(params, state, b, s + 1))
    t_cpu = time.time() - t0
    outs, tr, fb = _run_steps(m, batches, graphs=True, warmup=2)
    assert tr.ga is not None and tr.single_graph                     # steps 3 and 4 really replayed the ONE graph
    assert fb == {}, f"library fallbacks on the benched path: {fb}"
    rec = {"loss_rel_dev_vs_fp32_cpu_oracle": [], "image_embedding_rms_dev": [], "image_embedding_max_dev": [],
           "spot_embedding_max_dev": []}
    for s in range(steps):
        l, lr_ = outs[s]["loss"], float(ref[s]["loss"])
        d_i = outs[s]["image_embeddings"] - ref[s]["image_embeddings"]
        de_i, rms_i = float(d_i.abs().max()), float(d_i.pow(2).mean().sqrt())
        de_s = float((outs[s]["spot_embeddings"] - ref[s]["spot_embeddings"]).abs().max())
        rec["loss_rel_dev_vs_fp32_cpu_oracle"].append(round(abs(l - lr_) / max(1.0, abs(lr_)), 6))
        rec["image_embedding_rms_dev"].append(round(rms_i, 4))
        rec["image_embedding_max_dev"].append(round(de_i, 4))
        rec["spot_embedding_max_dev"].append(float(f"{de_s:.3e}"))
        print(f"cfg1 step {s + 1}: loss {l:.5f} oracle {lr_:.5f} (rel {abs(l - lr_) / max(1.0, abs(lr_)):.2e}); "
              f"dE_img max {de_i:.3e} rms {rms_i:.3e}; max|dE_spot| {de_s:.3e}")
        assert abs(l - lr_) <= 3e-2 * max(1.0, abs(lr_)), (s, l, lr_)
        assert de_i <= 1.5 and rms_i <= 0.25, (s, de_i, rms_i)
        assert de_s <= (2e-3, 0.071, 0.092, 0.142)[s], (s, de_s)
    worst, name = _worst_param_diff(m, params)
    print(f"cfg1: worst non-backbone parameter deviation after {steps} Adam steps {worst:.3e} ({name}); "
          f"oracle {t_cpu / steps:.1f} s/step")
    assert worst <= 8.5e-4, (worst, name)
    # the measured figures of THIS run as an artifact (bench.py relays profiles/parity_at_benched_shape.json; ADVICE r04: no
    # hard-coded parity constants in the bench line).  Written under gpurun_out/ (scratch); copied into profiles/ by hand.
    try:
        import json, os, subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        rec.update({"steps": steps, "worst_non_backbone_param_dev_after_4_adam_steps": float(f"{worst:.3e}"),
                    "measured_by": "tests/test_configs_gpu.py::test_cfg1_as_benched_vs_oracle (bf16 backbone kernels, fused "
                                   "InfoNCE, one HIP graph incl. FusedAdam: the mode bench.py times) vs oracle/ref_cpu.py"})
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_at_benched_shape.json"), "w") as f:
            json.dump(rec, f, indent=1)
    except OSError:
        pass


# ------------------------------------------------------------------------------------------------ configs[0]
@pytest.mark.parametrize("mode", ["fp32_exact", "bf16_fused_graph"])
def test_cfg0_her2st_shape_with_densenet(mode):
    """her2st config: B = 8, 112x112 patches, G = 785, full DenseNet-121, 3 steps vs the oracle.
    fp32_exact: fp32 backbone (round 4: im2col + exact-fp32 MFMA GEMM convolutions, own BN / pooling kernels -- no library call),
    exact InfoNCE -- 1e-4 on the loss at step 1 (north_star), 5e-3 afterwards (Adam's sign-like first updates amplify the
    summation-order noise of two fp32 executions of this 121-layer random-init net);
    bf16_fused_graph: the benched mode at this shape -- 10 % on the loss (measured 5 %: with 8 patches the last block's
    BatchNorm layers normalise over 8 x 3 x 3 = 72 values, which amplifies bf16 rounding far more than at batch 128)."""
    from mclstexp_amd import synth
    B, G, HW, steps = 8, 785, 112, 3
    bf16 = mode != "fp32_exact"
    m, params = _model_and_oracle_params(G, backbone_dtype=torch.bfloat16 if bf16 else None,
                                         infonce="fused" if bf16 else "exact")
    batches = [synth.make_batch(B, G, image_hw=HW, seed=s) for s in range(steps)]
    state, ref = {}, []
    for s, b in enumerate(batches):
        ref.append(_oracle_step(params, state, b, s + 1))
    outs, tr, fb = _run_steps(m, batches, graphs=bf16, warmup=2)
    if bf16:
        assert tr.ga is not None and fb == {}, fb
    tol = 1e-1 if bf16 else 5e-3
    for s in range(steps):
        l, lr_ = outs[s]["loss"], float(ref[s]["loss"])
        d_i = outs[s]["image_embeddings"] - ref[s]["image_embeddings"]
        de_i, rms_i = float(d_i.abs().max()), float(d_i.pow(2).mean().sqrt())
        print(f"cfg0 {mode} step {s + 1}: loss {l:.5f} oracle {lr_:.5f}; dE_img max {de_i:.3e} rms {rms_i:.3e}")
        assert abs(l - lr_) <= tol * max(1.0, abs(lr_)), (mode, s, l, lr_)
        assert de_i <= (1.5 if bf16 else 0.1) and rms_i <= (0.25 if bf16 else 0.02), (mode, s, de_i, rms_i)
        if not bf16 and s == 0:
            # round 4: the fp32 mode runs im2col + the exact-fp32 MFMA GEMM (no MIOpen): same weights, same batch, before any
            # optimizer step -- the north-star bound (1e-4 on the loss) holds end to end through the 121-layer backbone
            # (measured 1e-5 on the loss, image embeddings 3.0e-5 max / 7.6e-6 rms)
            assert abs(l - lr_) <= 1e-4 and de_i <= 2e-4 and rms_i <= 5e-5, (l, lr_, de_i, rms_i)
    worst, name = _worst_param_diff(m, params)
    print(f"cfg0 {mode}: worst non-backbone parameter deviation {worst:.3e} ({name})")
    assert worst <= 6.5e-4, (worst, name)          # 3 steps x 2 lr (see test_cfg1_as_benched_vs_oracle)


# ------------------------------------------------------------------------------------------------ configs[4] backbone
def _encoder(seed=0):
    from mclstexp_amd.backbones import ImageEncoder
    torch.manual_seed(seed)
    enc = ImageEncoder()
    with torch.no_grad():
        for mod in enc.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.2, 0.2)
    return enc


def test_reference_train_loop_stays_on_hip_kernels():
    """/root/reference/train.py:36-39,118-120 verbatim -- ``loss = model(batch); optimizer.zero_grad(); loss.backward();
    optimizer.step()`` with the STOCK torch.optim.Adam (zero_grad sets every .grad to None) -- on the bf16 fused backbone:
    no library fallback may be taken (the backward kernels create the dense .grad buffers they accumulate into), every
    parameter receives a finite gradient, the loss decreases, and the gradients agree with those the FusedAdam flat-bucket
    path produces for the same weights and batch."""
    import copy
    from mclstexp_amd import densenet_fused as dn, synth
    from mclstexp_amd.model import mclSTExp_Attention
    from mclstexp_amd.optim import FusedAdam
    G = 171
    torch.manual_seed(0)
    base = mclSTExp_Attention("densenet121", 100.0, 1024, G, 256, 8, 64, 2, backbone_dtype=torch.bfloat16)
    sd = base.state_dict()
    sd.update(synth.make_params(G, 1024, seed=0))
    base.load_state_dict(sd)
    batch = {k: v.to(DEV) for k, v in synth.make_batch(6, G, image_hw=224, seed=0).items()}
    batch["image"] = batch["image"].contiguous(memory_format=torch.channels_last)

    m = copy.deepcopy(base).to(DEV).to(memory_format=torch.channels_last).train()
    optimizer = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-3)
    dn.reset_fallbacks()
    losses = []
    for _ in range(3):
        loss = m(batch)
        optimizer.zero_grad()
        loss.backward()
        if not losses:
            g_stock = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        optimizer.step()
        losses.append(float(loss.item()))
    assert dn.fallback_counts() == {}, f"the reference loop left the HIP kernels: {dn.fallback_counts()}"
    names = {n for n, _ in m.named_parameters()}
    assert set(g_stock) == names, sorted(names - set(g_stock))[:5]
    assert all(torch.isfinite(g).all() for g in g_stock.values())
    assert losses[-1] < losses[0], losses

    m2 = copy.deepcopy(base).to(DEV).to(memory_format=torch.channels_last).train()
    opt2 = FusedAdam(m2.parameters(), lr=1e-4, weight_decay=1e-3).attach_model(m2)
    l2 = m2(batch); opt2.zero_grad(); l2.backward()
    dn.set_weight_provider(None)
    assert abs(float(l2.item()) - losses[0]) <= 1e-5 * max(1.0, abs(losses[0])), (float(l2.item()), losses[0])
    worst = 0.0
    for n, p in m2.named_parameters():
        if n.startswith("image_encoder"):
            d = (p.grad - g_stock[n]).abs().max().item() / (g_stock[n].abs().max().item() + 1e-20)
            worst = max(worst, d)
    # same deterministic kernels, same operands: only the bf16 weight copies are produced differently (per-tensor cast vs flat
    # shadow), bit-identical values
    assert worst <= 1e-6, worst


def test_cfg4_backbone_256px_accuracy_vs_fp64():
    """256x256 patches (64-wide maps in block 1: the widest the 3x3 slab kernels see): the fused bf16 execution must be
    as close to an fp64 run of the same module as the stock bf16-autocast module path is (B = 4)."""
    from mclstexp_amd import densenet_fused as dn
    base = _encoder()
    ref64 = copy.deepcopy(base).double().to(DEV).train()
    ref = copy.deepcopy(base).to(DEV).train()
    fus = copy.deepcopy(base).to(DEV).to(memory_format=torch.channels_last).train()     # as train.py / bench.py hold it
    g = torch.Generator().manual_seed(3)
    x = torch.rand(4, 3, 256, 256, generator=g).to(DEV)
    dy = (torch.rand(4, 1024, generator=g) - 0.5).to(DEV)
    y64 = ref64(x.double()); y64.backward(dy.double())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y_ref = ref(x.contiguous(memory_format=torch.channels_last)).float()
    y_ref.backward(dy)
    for p in fus.parameters():                      # dense fp32 .grad buffers, as FusedAdam's flat bucket provides:
        p.grad = torch.zeros_like(p)                # the weight-gradient kernels accumulate straight into them
    dn.reset_fallbacks()
    y = fus.forward_fused(x, torch.bfloat16); y.backward(dy)
    assert dn.fallback_counts() == {}, dn.fallback_counts()
    scale = y64.abs().max().item()
    e_ref = (y_ref.double() - y64).abs().max().item() / scale
    e_fus = (y.double() - y64).abs().max().item() / scale
    d_ref, d_fus = [], []
    for (n, p64), (_, p), (_, q) in zip(ref64.named_parameters(), ref.named_parameters(), fus.named_parameters()):
        s_ = p64.grad.abs().max().item() + 1e-30
        d_ref.append((p.grad.double() - p64.grad).abs().max().item() / s_)
        d_fus.append((q.grad.double() - p64.grad).abs().max().item() / s_)
    d_ref, d_fus = np.array(d_ref), np.array(d_fus)
    print(f"256px: features err stock-bf16 {e_ref:.2e} fused {e_fus:.2e}; grad rel-dev median stock {np.median(d_ref):.2e} "
          f"fused {np.median(d_fus):.2e}; max stock {d_ref.max():.2e} fused {d_fus.max():.2e}")
    assert e_fus <= 2.0 * e_ref + 1e-5, (e_fus, e_ref)
    assert np.median(d_fus) <= 2.0 * np.median(d_ref) + 1e-5
    assert d_fus.max() <= 2.5 * d_ref.max() + 1e-4


def test_cfg4_backbone_full_batch_properties():
    """B = 256 x 256x256 (configs[4]'s per-GPU batch): size-independent properties of the train-mode forward/backward --
    no library fallback, finite, batch-permutation equivariance (BatchNorm statistics are permutation-invariant, so
    permuting the batch permutes features and leaves every parameter gradient unchanged, up to bf16 re-rounding of the
    re-ordered sums), and bit-reproducibility run to run (no atomics anywhere on the path)."""
    from mclstexp_amd import densenet_fused as dn
    enc = _encoder().to(DEV).to(memory_format=torch.channels_last).train()
    g = torch.Generator(device=DEV).manual_seed(11)
    B = 256
    x = torch.rand(B, 3, 256, 256, device=DEV, generator=g)
    dy = torch.rand(B, 1024, device=DEV, generator=g) - 0.5
    perm = torch.randperm(B, device=DEV, generator=g)

    def run(xx, dd):
        for p in enc.parameters():                  # dense fp32 .grad buffers (FusedAdam's flat bucket provides them):
            if p.grad is None:                      # the weight-gradient kernels accumulate straight into them
                p.grad = torch.zeros_like(p)
            p.grad.zero_()
        y = enc.forward_fused(xx, torch.bfloat16)
        y.backward(dd)
        return y.detach(), {n: p.grad.detach().clone() for n, p in enc.named_parameters()}

    dn.reset_fallbacks()
    y1, g1 = run(x, dy)
    assert dn.fallback_counts() == {}, dn.fallback_counts()
    assert torch.isfinite(y1).all() and all(torch.isfinite(v).all() for v in g1.values())
    y1b, g1b = run(x, dy)
    assert torch.equal(y1, y1b)
    for n in g1:
        assert torch.equal(g1[n], g1b[n]), f"{n}: not bit-reproducible run to run"
    y2, g2 = run(x[perm].contiguous(), dy[perm].contiguous())
    err = (y2 - y1[perm]).abs().max().item() / y1.abs().max().item()
    assert err <= 8e-2, err                       # measured 4e-2: bf16 re-rounding of the re-ordered BatchNorm sums
    devs = np.array([((g2[n] - g1[n]).abs().max() / (g1[n].abs().max() + 1e-30)).item() for n in g1])
    # Parameter gradients of this random-init net under a random upstream gradient are sums of almost cancelling terms:
    # ANY bf16 execution moves them by tens of percent when the summation order changes.  The yardstick is therefore the
    # stock bf16-autocast module path put through the same permutation: the fused path must be as permutation-stable.
    ref = copy.deepcopy(enc)
    for p in ref.parameters():
        p.grad = None

    def run_ref(xx, dd):
        for p in ref.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = ref(xx.contiguous(memory_format=torch.channels_last)).float()
        y.backward(dd)
        return {n: p.grad.detach().clone() for n, p in ref.named_parameters()}

    r1, r2 = run_ref(x, dy), run_ref(x[perm].contiguous(), dy[perm].contiguous())
    devs_ref = np.array([((r2[n] - r1[n]).abs().max() / (r1[n].abs().max() + 1e-30)).item() for n in r1])
    print(f"B=256 256px: permutation feature dev {err:.2e}; grad dev median fused {np.median(devs):.2e} "
          f"stock {np.median(devs_ref):.2e}; max fused {devs.max():.2e} stock {devs_ref.max():.2e}")
    assert np.median(devs) <= 2.0 * np.median(devs_ref) + 1e-3, (np.median(devs), np.median(devs_ref))


# ------------------------------------------------------------------------------------------------ configs[2]
@pytest.mark.parametrize("name", ["vit_base_patch16_224", "vit_base_patch32_224"])
def test_cfg2_vit_b256_vs_fp64(name):
    """ViT-B at batch 256 (configs[2] says /16; the reference's default is /32, model.py:106): the GPU execution the
    model uses for this encoder (bf16) against an fp64 run of the same module -- features 3e-2 of max|feature|,
    parameter gradients: median relative deviation 3e-2 (parity of the architecture itself is unpinned: timm absent)."""
    from mclstexp_amd.backbones import ImageEncoder_VIT
    from mclstexp_amd.model import mclSTExp_Attention
    torch.manual_seed(0)
    enc = ImageEncoder_VIT(name)
    B = 256
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, 224, 224, generator=g).to(DEV)
    dy = (torch.rand(B, 768, generator=g) - 0.5).to(DEV)
    ref64 = copy.deepcopy(enc).double().to(DEV).train()
    # fp64 reference in chunks of the batch (no cross-sample coupling in a ViT: LayerNorm only)
    g64 = None
    y64 = []
    for i in range(0, B, 64):
        yy = ref64(x[i:i + 64].double())
        yy.backward(dy[i:i + 64].double())
        y64.append(yy.detach())
    y64 = torch.cat(y64)
    m = mclSTExp_Attention("identity", 1.0, 768, 171, 256, 8, 64, 1, backbone_dtype=torch.bfloat16)
    m.image_encoder = copy.deepcopy(enc)
    m.to(DEV).train()
    y = m._encode_image(m.image_encoder, x)
    y.backward(dy)
    scale = y64.abs().max().item()
    err = (y.double() - y64).abs().max().item() / scale
    devs = []
    for (n, p64), (_, q) in zip(ref64.named_parameters(), m.image_encoder.named_parameters()):
        assert q.grad is not None, n
        devs.append(((q.grad.double() - p64.grad).abs().max() / (p64.grad.abs().max() + 1e-30)).item())
    devs = np.array(devs)
    print(f"{name} B=256: feature err {err:.2e}; grad rel-dev median {np.median(devs):.2e} max {devs.max():.2e}")
    assert err <= 3e-2, err
    assert np.median(devs) <= 3e-2 and devs.max() <= 0.2, (np.median(devs), devs.max())

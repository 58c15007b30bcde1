"""Pins the CPU oracle (oracle/ref_cpu.py) to fixtures captured from the reference's own
classes (tests/golden/gen_goldens.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from mclstexp_amd import synth
from oracle import ref_cpu
from helpers import (GOLDEN_CASES, UNTOUCHED_ROW, assert_close, assert_close_scaled, layer_sample, load_golden,
                     oracle_forward, sample)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_oracle_matches_reference_fixture(name):
    z, meta = load_golden(name)
    B, G, D, L = meta["B"], meta["G"], meta["D"], meta["layers"]
    params = synth.make_params(G, D, 256, 8, 64, L, seed=0)
    for p in params.values():
        p.requires_grad_(True)
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in params.items()}
    for step in range(meta["steps"]):
        batch = synth.make_batch(B, G, image_dim=D, seed=step)
        out = oracle_forward(params, batch, meta)
        s = out["cos_smi"]
        s.retain_grad()
        for p in params.values():
            p.grad = None
        out["loss"].backward()
        # 1e-4 absolute on loss and logits is north_star's bar
        assert_close(out["loss"].item(), z[f"step{step}.loss"], 1e-4, what=f"{name} loss step {step}")
        if step == 0:
            assert_close(s.detach().numpy(), z["cos_smi"], 5e-5, what="cos_smi")
            assert_close(out["image_embeddings"].detach().numpy(), z["image_embeddings"], 1e-5, what="image_embeddings")
            assert_close(out["spot_embeddings"].detach().numpy(), z["spot_embeddings"], 1e-5, what="spot_embeddings")
            assert_close(s.grad.numpy(), z["dS"], 1e-6, what="dS (autograd)")
            assert_close(ref_cpu.symmetric_infonce_grad(s.detach()).numpy(), z["dS"], 1e-6, what="dS (closed form)")
            for l in range(L):
                full = out["layer_outs"][l].detach()
                got = layer_sample(full, B, G).numpy()
                assert_close(got, z[f"layer{l}_out"], 2e-5, what=f"layer{l}_out")
            for n, p in params.items():
                if n in ("x_embed.weight", "y_embed.weight"):
                    rows = torch.from_numpy(z["grad_rows." + n])
                    got = p.grad[rows][:, :: max(1, G // 64)].numpy()
                    assert_close(got, z["grad." + n], 1e-6, 1e-5, what="grad " + n)
                    assert float(p.grad[UNTOUCHED_ROW].abs().max()) == 0.0
                else:
                    ref = z["grad." + n]
                    scale = float(np.abs(ref).max()) + 1e-12
                    assert_close(sample(p.grad), ref, 2e-5 * scale + 1e-7, what="grad " + n)
        # torch.optim.Adam(lr=1e-4, weight_decay=1e-3) restated
        with torch.no_grad():
            for n, p in params.items():
                m, v = state[n]
                ref_cpu.adam_l2_step(p, p.grad, m, v, step + 1)
        tag = f"step{step}."
        # later steps inherit sign(g)-amplified fp32 noise from Adam's first update (g/(|g|+eps))
        rm, rv = (2e-5, 4e-5) if step == 0 else (1e-3, 1e-3)
        for n, p in params.items():
            m, v = state[n]
            if n in ("x_embed.weight", "y_embed.weight"):
                rows = torch.from_numpy(z[tag + "rows." + n])
                cs = slice(None, None, max(1, G // 64))
                assert_close(p.detach()[rows][:, cs].numpy(), z[tag + "param." + n], 2e-6 if step == 0 else 2e-5, what=tag + "param " + n)
                assert_close_scaled(m[rows][:, cs].numpy(), z[tag + "exp_avg." + n], rm, what=tag + "m " + n)
                assert_close_scaled(v[rows][:, cs].numpy(), z[tag + "exp_avg_sq." + n], rv, what=tag + "v " + n)
            else:
                assert_close(sample(p), z[tag + "param." + n], 2e-6 if step == 0 else 2e-5, what=tag + "param " + n)
                assert_close_scaled(sample(m), z[tag + "exp_avg." + n], rm, what=tag + "m " + n)
                assert_close_scaled(sample(v), z[tag + "exp_avg_sq." + n], rv, what=tag + "v " + n)


def test_infonce_closed_form_equals_soft_label_ce():
    """model.py:243-247 (float identity soft labels) == the closed form the kernels implement."""
    torch.manual_seed(1)
    s = torch.randn(37, 37) * 30.0
    eye = torch.eye(37)
    F = torch.nn.functional
    ref = (F.cross_entropy(s, eye) + F.cross_entropy(s.t(), eye.t())) / 2.0
    assert abs(ref.item() - ref_cpu.symmetric_infonce(s).item()) < 1e-5


@pytest.mark.parametrize("world", [2, 4])
def test_infonce_strip_sums_to_global(world):
    """DP decomposition: per-rank row/col strips reproduce the global loss."""
    torch.manual_seed(2)
    b_loc, P = 5, 32
    es = torch.randn(world * b_loc, P)
    ei = torch.randn(world * b_loc, P)
    glob = ref_cpu.symmetric_infonce(ref_cpu.logits(es, ei, 0.7))
    tot = 0.0
    for r in range(world):
        sl = slice(r * b_loc, (r + 1) * b_loc)
        rp, cp, _ = ref_cpu.infonce_strip(es[sl], ei[sl], es, ei, r * b_loc, 0.7)
        tot += rp.item() + cp.item()
    assert abs(tot / (2 * world * b_loc) - glob.item()) < 1e-5


def test_densenet_restatement_shapes():
    """Backbone restatement (parity unpinned): shape/finite check at a tiny size."""
    from mclstexp_amd.backbones import densenet121_features_module
    torch.manual_seed(0)
    net = densenet121_features_module()
    p = {"image_encoder.model.0." + k: v for k, v in net.state_dict().items()}
    x = torch.rand(2, 3, 64, 64)
    y = ref_cpu.densenet121_features(p, x)
    assert y.shape == (2, 1024) and torch.isfinite(y).all()
    net.train()
    y2 = torch.nn.functional.adaptive_avg_pool2d(net(x), (1, 1)).flatten(1)
    assert_close(y.detach().numpy(), y2.detach().numpy(), 1e-4, what="densenet module vs functional")


# --------------------------------------------------------------------------- retrieval
from oracle import ref_retrieval  # noqa: E402
from helpers import RETRIEVAL_CASES, check_topk, load_retrieval_golden  # noqa: E402


@pytest.mark.parametrize("name", sorted(RETRIEVAL_CASES))
def test_retrieval_oracle_matches_reference_fixture(name):
    """oracle/ref_retrieval.py against the outputs of the reference's own find_matches / weighting loops."""
    z, case, meta = load_retrieval_golden(name)
    values, indices = ref_retrieval.find_matches(case["spot_key"], case["image_query"], top_k=meta["top_k"])
    sim64 = ref_retrieval.similarity_f64(case["spot_key"], case["image_query"])
    # same torch build, same ops: expected to agree exactly; the tolerance-aware check documents what
    # "equal" means should a BLAS change reorder the fp32 sums
    check_topk(indices, sim64, values=values, what=f"{name} oracle")
    check_topk(z["indices"], sim64, what=f"{name} reference")
    assert (indices == z["indices"]).mean() > 0.999
    if "values" in z.files:
        assert_close(values, z["values"], 1e-6, what="values")
    # the weighting loop on the REFERENCE's indices: pure numpy, must agree to rounding
    emb, expr = ref_retrieval.weighted_prediction(case["spot_key"], case["expression_key"], case["image_query"],
                                                  z["indices"], ord=meta["ord"])
    assert_close(emb, z["emb_pred"], 1e-12, 1e-12, what="emb_pred")
    assert_close(expr, z["expr_pred"], 1e-12, 1e-12, what="expr_pred")


# --------------------------------------------------------------------------- BLEEP soft-target loss
from helpers import BLEEP_CASES, bleep_embeddings  # noqa: E402


@pytest.mark.parametrize("name", BLEEP_CASES)
def test_bleep_loss_oracle_matches_reference_fixture(name):
    """oracle/ref_cpu.bleep_soft_clip_loss against loss and autograd gradients of the reference's own loss section."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bleep_loss.npz"))
    B, seed, vit = [int(v) for v in z[name + ".meta"]]
    es, ei = bleep_embeddings(B, seed)
    es.requires_grad_(True); ei.requires_grad_(True)
    loss = ref_cpu.bleep_soft_clip_loss(es, ei, float(z[name + ".T"]), bool(vit))
    loss.backward()
    assert_close(loss.item(), z[name + ".loss"], 1e-6, what="loss")
    assert_close(es.grad.numpy(), z[name + ".d_es"], 1e-7, 1e-5, what="d_es")
    assert_close(ei.grad.numpy(), z[name + ".d_ei"], 1e-7, 1e-5, what="d_ei")


# --------------------------------------------------------------------------- input pipeline
from helpers import INPUT_CASE, synthetic_slide  # noqa: E402


def test_input_oracle_matches_pil_fixture():
    """oracle/ref_input.py against PIL's own crop / ToTensor / flips / quarter-turn rotations (bit-exact)."""
    from oracle import ref_input
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "input_patches.npz"))
    img = synthetic_slide()
    r = INPUT_CASE["r"]
    got = np.stack([ref_input.to_tensor(ref_input.crop(img, y, x, r)) for (x, y) in INPUT_CASE["centers_xy"]])
    assert np.array_equal(got, z["eval"])
    got = np.stack([ref_input.tenx_transform(ref_input.crop(img, v1, v2, r), bool(h), bool(v), a).transpose(2, 0, 1)
                    .astype(np.float32) for (v1, v2), h, v, a in zip(INPUT_CASE["tenx_centers"], INPUT_CASE["hflip"],
                                                                     INPUT_CASE["vflip"], INPUT_CASE["angle"])])
    assert np.array_equal(got, z["tenx"])
    # normalisation: rows sum to 1e4 before the log; empty spots stay zero
    c = np.array([[1.0, 3.0, 0.0, 6.0], [0.0, 0.0, 0.0, 0.0]])
    y = ref_input.log_library_size_normalize(c)
    assert_close(y[0], np.log10(c[0] * 1000.0 + 1.0), 1e-12, what="log10(x / libsize * 1e4 + 1)")
    assert (y[1] == 0).all()


def test_train_augmentation_oracle_matches_pil_fixture():
    """oracle/ref_input.her2st_train_transform (ColorJitter / flip / arbitrary-angle nearest rotation restated from PIL's
    ImageEnhance, Blend.c and Geometry.c arithmetic) against PIL's own outputs for explicit draws: bit-exact."""
    from oracle import ref_input
    from helpers import AUG_CASE
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "input_augment.npz"))
    img = synthetic_slide()
    r = AUG_CASE["r"]
    got = np.stack([ref_input.her2st_train_transform(ref_input.crop(img, y, x, r), AUG_CASE["order"][i],
                                                     AUG_CASE["brightness"][i], AUG_CASE["contrast"][i],
                                                     AUG_CASE["saturation"][i], bool(AUG_CASE["hflip"][i]),
                                                     AUG_CASE["angle"][i])
                    for i, (x, y) in enumerate(AUG_CASE["centers_xy"])])
    assert np.array_equal(got, z["train"])

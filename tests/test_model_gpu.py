"""End-to-end parity of the drop-in model on the MI355X against the committed reference fixtures
(tests/golden, produced by the reference's own classes) -- loss and logits within 1e-4 (north_star),
gradients, and Adam trajectories with both the stock torch optimizer and FusedAdam."""
import numpy as np
import pytest
import torch

from helpers import (GOLDEN_CASES, UNTOUCHED_ROW, assert_close, assert_close_scaled, load_golden, sample)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(meta, embedding_grad="dense"):
    from mclstexp_amd import synth
    from mclstexp_amd.model import mclSTExp_Attention, mclSTExp_MLP
    G, D, L = meta["G"], meta["D"], meta["layers"]
    if meta["variant"] == "mlp":
        m = mclSTExp_MLP(meta["T"], D, G, 256, encoder_name="identity", embedding_grad=embedding_grad)
    else:
        m = mclSTExp_Attention("identity", meta["T"], D, G, 256, 8, 64, L, embedding_grad=embedding_grad)
    m.load_state_dict(synth.make_params(G, D, 256, 8, 64, L, seed=0), strict=True)
    m.to(DEV).train()
    m.capture = True
    return m


def to_dev(batch):
    return {k: v.to(DEV) for k, v in batch.items()}


@pytest.mark.parametrize("name", GOLDEN_CASES)
@pytest.mark.parametrize("optimizer", ["torch_adam_dense", "fused_adam_rowsparse", "fused_adam_rowsparse_dense_tables"])
def test_model_matches_reference_fixture(name, optimizer):
    from mclstexp_amd import synth
    from mclstexp_amd.optim import FusedAdam
    z, meta = load_golden(name)
    B, G, D, L = meta["B"], meta["G"], meta["D"], meta["layers"]
    fused = optimizer.startswith("fused")
    m = build(meta, "rowsparse" if fused else "dense")
    if fused:
        # default: lazy-exact position tables (untouched rows replayed on demand); "dense_tables": every row on every step
        lazy = not optimizer.endswith("dense_tables")
        opt = FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3, lazy_tables=lazy).attach_model(m)
    else:
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-3)     # train.py:118-120, unchanged
    named = dict(m.named_parameters())
    for step in range(meta["steps"]):
        batch = to_dev(synth.make_batch(B, G, image_dim=D, seed=step))
        loss = m(batch)
        opt.zero_grad()
        loss.backward()
        assert_close(loss.item(), z[f"step{step}.loss"], 1e-4, what=f"loss step {step} (1e-4 abs)")
        if step == 0:
            assert_close(m.last["logits"].cpu(), z["cos_smi"], 1e-4, what="cos_smi (1e-4 abs)")
            assert_close(m.last["image_embeddings"].cpu(), z["image_embeddings"], 2e-5, what="image_embeddings")
            assert_close(m.last["spot_embeddings"].cpu(), z["spot_embeddings"], 2e-5, what="spot_embeddings")
            for n, p in named.items():
                if n in ("x_embed.weight", "y_embed.weight"):
                    if fused:
                        assert p.grad is None       # dense (65536, G) gradient never materialised
                        continue
                    rows = torch.from_numpy(z["grad_rows." + n]).to(DEV)
                    got = p.grad[rows][:, :: max(1, G // 64)].cpu()
                    assert_close_scaled(got, z["grad." + n], 2e-5, what="grad " + n)
                    assert float(p.grad[UNTOUCHED_ROW].abs().max()) == 0.0
                else:
                    assert_close_scaled(sample(p.grad), z["grad." + n], 3e-5, what="grad " + n)
                    assert abs(p.grad.double().sum().item() - float(z["gradsum." + n])) <= \
                        3e-5 * float(z["gradabs." + n]) + 1e-7, "gradsum " + n
        opt.step()
        tag = f"step{step}."
        # Adam's first update is lr * g / (|g| + eps): where |g| ~ eps = 1e-8 a gradient difference d moves the parameter by up to
        # lr / eps * d = 1e4 d, so the summation-order noise of a gradient (a few 1e-10 here) shows as a few 1e-6 in one or two
        # of the position-table elements (2.3e-6 seen with the single-launch projection head's other summation order)
        tol_p = 4e-6 if step == 0 else 3e-5
        last = step == meta["steps"] - 1
        if fused and lazy and last:
            opt.materialize_tables()    # the fixture's untouched row (60000) has been replayed over all steps at once
        for n, p in named.items():
            if n in ("x_embed.weight", "y_embed.weight"):
                if fused and lazy and not last:
                    continue            # rows no batch has touched are behind until materialised (optim.py)
                rows = torch.from_numpy(z[tag + "rows." + n]).to(DEV)
                cs = slice(None, None, max(1, G // 64))
                assert_close(p.detach()[rows][:, cs].cpu(), z[tag + "param." + n], tol_p, what=tag + "param " + n)
            else:
                assert_close(sample(p), z[tag + "param." + n], tol_p, what=tag + "param " + n)


def test_eval_submodule_calls_like_reference_eval_script():
    """evel_her2st.py:48-69 calls the sub-modules piecewise under no_grad/eval."""
    from mclstexp_amd import synth
    from oracle import ref_cpu
    z, meta = load_golden("b33_g171")
    m = build(meta).eval()
    batch = synth.make_batch(meta["B"], meta["G"], image_dim=meta["D"], seed=0)
    with torch.no_grad():
        img = m.image_projection(m.image_encoder(batch["image"].to(DEV)))
        x = batch["position"][:, 0].long().to(DEV)
        y = batch["position"][:, 1].long().to(DEV)
        feat = batch["expression"].to(DEV) + m.x_embed(x) + m.y_embed(y)
        spot = m.spot_projection(m.spot_encoder(feat.unsqueeze(dim=0))).squeeze(dim=0)
    assert_close(img.cpu(), z["image_embeddings"], 2e-5, what="eval image path")
    assert_close(spot.cpu(), z["spot_embeddings"], 2e-5, what="eval spot path")


def test_unfused_module_path_matches_fused():
    """PreNorm/Attention/FeedForward called individually == attn_block's fused path."""
    from mclstexp_amd import synth
    z, meta = load_golden("b33_g171")
    m = build(meta)
    x = synth.make_batch(meta["B"], meta["G"], image_dim=meta["D"], seed=0)["expression"].to(DEV)
    blk = m.spot_encoder[0]
    x1 = x.clone().requires_grad_(True)
    y_fused = blk(x1.unsqueeze(0)).squeeze(0)
    y_fused.sum().backward()
    g_fused = {n: p.grad.clone() for n, p in blk.named_parameters()}
    for p in blk.parameters():
        p.grad = None
    x2 = x.clone().requires_grad_(True)
    t = blk.attn(x2) + x2
    y_un = blk.ff(t) + t
    y_un.sum().backward()
    assert_close(y_un.detach().cpu(), y_fused.detach().cpu(), 2e-5, what="unfused vs fused forward")
    assert_close_scaled(x2.grad.cpu(), x1.grad.cpu(), 2e-5, what="unfused vs fused dx")
    for n, p in blk.named_parameters():
        assert_close_scaled(p.grad.cpu(), g_fused[n].cpu(), 3e-5, what="unfused vs fused grad " + n)


def test_cpu_input_raises():
    from mclstexp_amd import synth
    z, meta = load_golden("b8_g785")
    m = build(meta)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(synth.make_batch(8, 785, image_dim=1024, seed=0))


def test_densenet_backbone_end_to_end_small():
    """Full model incl. the delegated DenseNet-121 (fp32) vs the oracle's functional restatement on CPU
    (backbone parity unpinned: restatement vs restatement), B=4, 64x64 patches."""
    from mclstexp_amd import synth
    from mclstexp_amd.model import mclSTExp_Attention
    from oracle import ref_cpu
    torch.manual_seed(0)
    B, G = 4, 171
    m = mclSTExp_Attention("densenet121", 1.0, 1024, G, 256, 8, 64, 2)
    sd = m.state_dict()
    sd.update(synth.make_params(G, 1024, seed=0))
    m.load_state_dict(sd)
    batch = synth.make_batch(B, G, image_hw=64, seed=0)
    params = {k: v.clone() for k, v in m.state_dict().items()}
    feats = ref_cpu.densenet121_features(params, batch["image"])
    ref = ref_cpu.forward_from_features(params, feats, batch["expression"], batch["position"], 1.0, 2, 8, 64)
    m.to(DEV).train()
    loss = m(to_dev(batch))
    loss.backward()
    assert abs(loss.item() - ref["loss"].item()) < 5e-3, (loss.item(), ref["loss"].item())
    assert m.image_encoder.model[0].conv0.weight.grad is not None

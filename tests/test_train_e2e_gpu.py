"""End to end on the MI355X: train.py's loop (HIP-graph replay, FusedAdam) -> checkpoint in the reference's wire format
-> reload with the reference's key rewrites -> eval-mode embedding extraction -> retrieval.  """
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_save_load_embed_retrieve(tmp_path):
    from mclstexp_amd import retrieval, synth, train
    from mclstexp_amd.model import load_reference_state_dict, mclSTExp_Attention
    argv = ["--batch_size", "16", "--dim", "171", "--image_size", "64", "--max_epochs", "2", "--steps_per_epoch", "6",
            "--hip_graphs", "--infonce", "fused", "--save_dir", str(tmp_path), "--log_every", "1"]
    train.main(argv)
    path = os.path.join(str(tmp_path), "her2st", "best_0.pt")
    sd = torch.load(path, map_location="cpu")
    assert "x_embed.weight" in sd and "image_encoder.model.0.denseblock4.denselayer16.conv2.weight" in sd
    assert all(torch.isfinite(v.float()).all() for v in sd.values())
    assert int(sd["image_encoder.model.0.norm0.num_batches_tracked"]) == 12          # BN running statistics were kept
    # the reference's eval scripts load with 'module.' stripped and 'well' -> 'spot' (evel_her2st.py:33-37)
    m = mclSTExp_Attention("densenet121", 1.0, 1024, 171, 256, 8, 64, 2, backbone_dtype=torch.bfloat16)
    load_reference_state_dict(m, {("module." + k).replace("spot", "well"): v for k, v in sd.items()})
    m.to("cuda").to(memory_format=torch.channels_last)
    loader = [synth.make_batch(16, 171, image_hw=64, seed=s) for s in range(3)]
    img, spot = retrieval.get_embeddings(m, loader)
    assert img.shape == spot.shape == (48, 256) and torch.isfinite(img).all() and torch.isfinite(spot).all()
    # trained pairs: an image embedding retrieves spot embeddings, and the weighted expression prediction is finite
    expr = np.concatenate([b["expression"].numpy() for b in loader])
    out = retrieval.predict_expression(spot, expr, img, top_k=5, ord=2)
    assert out["indices"].shape == (48, 5) and np.isfinite(out["matched_spot_expression_pred"]).all()

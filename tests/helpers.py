"""Shared test helpers: golden loading, procedural cases, oracle driving (tests only)."""
import os

import numpy as np
import torch

from mclstexp_amd import synth
from oracle import ref_cpu

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = ["b8_g785", "b33_g171", "b128_g1000", "b16_g685_vit", "b8_g171_mlp", "b256_g3467"]
UNTOUCHED_ROW = 60000


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    B, G, D, L, steps = [int(v) for v in z["meta"]]
    return z, dict(B=B, G=G, D=D, layers=L, steps=steps, T=float(z["temperature"]),
                   variant="mlp" if name.endswith("mlp") else "attention")


def layer_sample(v, B, G):
    """Sub-sampling of a per-layer (B, G) output as stored by tests/golden/gen_goldens.py."""
    return v if B * G <= 40000 else (v[::8] if B * G <= 200000 else v[::32, ::4])


def sample(t, n=256):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].cpu().numpy()


def oracle_forward(params, batch, meta):
    if meta["variant"] == "mlp":
        return ref_cpu.forward_mlp_from_features(params, batch["image"], batch["expression"],
                                                 batch["position"], meta["T"])
    return ref_cpu.forward_from_features(params, batch["image"], batch["expression"], batch["position"],
                                         meta["T"], meta["layers"], 8, 64)


def assert_close(a, b, atol, rtol=0.0, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), (f"{what}: max abs err {err.max():.3e} (tol {atol:g}+{rtol:g}*|ref|), "
                           f"{int(bad.sum())}/{bad.size} out of tolerance")


def assert_close_scaled(a, b, rel=1e-5, floor=1e-12, what=""):
    """|a-b| <= rel * max|b| + floor, elementwise (fp32 accumulation-order noise scales with the
    tensor's magnitude, not with each element's)."""
    b64 = np.asarray(b, dtype=np.float64)
    assert_close(a, b, rel * float(np.abs(b64).max() if b64.size else 0.0) + floor, what=what)


# --------------------------------------------------------------------------- retrieval
RETRIEVAL_CASES = {"her2st": 1, "cscc": 2, "visium": 2, "small_k1": 2}   # name -> ord of the distance norm


def load_retrieval_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"retrieval_{name}.npz"))
    N, Q, G, k, seed, dup = [int(v) for v in z["meta"]]
    case = synth.make_retrieval_case(N, Q, 256, G, seed=seed, duplicates=dup)
    return z, case, dict(N=N, Q=Q, G=G, top_k=k, ord=RETRIEVAL_CASES[name])


def check_topk(indices, sim64, tol=2e-6, values=None, what="topk"):
    """Tolerance-aware top-k check against float64 similarities ``sim64`` (Q,N).

    Exact wherever the float64 values are separated by more than ``tol`` (fp32 products and sums of 256
    terms differ between implementations by ~1e-7, so elements closer than that to the k-th value or to
    each other may legitimately swap):  every selected element is >= k-th value - tol, every element
    > k-th value + tol is selected, rows are sorted best-first within tol, no index repeats.
    Returns the number of rows whose index SET equals the float64 one exactly."""
    indices = np.asarray(indices)
    q, k = indices.shape
    exact_rows = 0
    for i in range(q):
        row = sim64[i]
        idx = indices[i]
        assert len(set(idx.tolist())) == k, f"{what}: row {i} repeats an index"
        assert idx.min() >= 0 and idx.max() < row.shape[0], f"{what}: row {i} index out of range"
        order = np.argsort(-row, kind="stable")
        kth = row[order[k - 1]]
        got = row[idx]
        assert got.min() >= kth - tol, f"{what}: row {i} selected {got.min():.9f} < k-th {kth:.9f}"
        must = np.nonzero(row > kth + tol)[0]
        missing = np.setdiff1d(must, idx)
        assert missing.size == 0, f"{what}: row {i} misses {missing[:5]} (clearly above the k-th value)"
        assert (np.diff(got) <= tol).all(), f"{what}: row {i} not sorted best-first"
        if values is not None:
            assert np.abs(np.asarray(values[i], dtype=np.float64) - got).max() <= tol, f"{what}: row {i} values"
        exact_rows += int(set(order[:k].tolist()) == set(idx.tolist()))
    return exact_rows


# --------------------------------------------------------------------------- BLEEP soft-target CLIP loss
BLEEP_CASES = ["clip_b8", "clip_b33_t07", "vit_b16_t05"]


def bleep_embeddings(B, seed):
    """Procedural (B, 256) spot / image embeddings: LayerNorm-ed, correlated pairs, small norm (so that the soft targets
    are not saturated)."""
    es = synth.uniform_tensor("bleep.es", (B, 256), -1.0, 1.0, seed)
    ei = synth.uniform_tensor("bleep.ei", (B, 256), -1.0, 1.0, seed)
    ei = 0.6 * es + 0.4 * ei
    ln = torch.nn.functional.layer_norm
    return (ln(es, (256,)) * 0.08).contiguous(), (ln(ei, (256,)) * 0.08).contiguous()


# --------------------------------------------------------------------------- input pipeline
INPUT_CASE = dict(r=16,
                  centers_xy=[(40, 50), (16, 16), (3, 70), (115, 95), (60, 2), (119, 99)],      # (x, y); some cross the border
                  tenx_centers=[(50, 40), (30, 60), (70, 80), (16, 16), (84, 104), (45, 45), (33, 77), (60, 30)],
                  hflip=[0, 1, 0, 1, 0, 1, 1, 0], vflip=[0, 0, 1, 1, 0, 0, 1, 1],
                  angle=[0, 90, 180, -90, 90, 180, -90, 0])


def synthetic_slide(hs=100, ws=120):
    """Procedural (hs, ws, 3) uint8 'whole-slide image'."""
    u = synth.uniform_tensor("slide", (hs, ws, 3), 0.0, 256.0, 0).numpy()
    return np.clip(np.floor(u), 0, 255).astype(np.uint8)


# HER2ST / cSCC training transform (dataset.py:63-68): explicit draws for the golden case (8 patches of 2r = 32 around
# centres of the synthetic slide, some crossing its border; quarter-turn and generic angles; factors below, at and above 1)
AUG_CASE = dict(r=16,
                centers_xy=[(40, 50), (16, 16), (3, 70), (115, 95), (60, 2), (119, 99), (50, 50), (70, 30)],
                order=[(0, 1, 2), (2, 1, 0), (1, 0, 2), (0, 2, 1), (2, 0, 1), (1, 2, 0), (0, 1, 2), (1, 2, 0)],
                brightness=[0.5, 1.5, 1.0, 0.73, 1.21, 0.99, 1.37, 0.61],
                contrast=[1.5, 0.5, 1.0, 1.18, 0.66, 1.45, 0.83, 1.02],
                saturation=[1.0, 1.49, 0.51, 0.9, 1.3, 0.7, 1.11, 0.58],
                hflip=[0, 1, 0, 1, 1, 0, 1, 0],
                angle=[0.0, 37.5, -90.0, 180.0, -133.7, 90.0, 12.25, -179.99])

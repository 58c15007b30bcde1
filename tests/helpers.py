"""Shared test helpers: golden loading, procedural cases, oracle driving (tests only)."""
import os

import numpy as np
import torch

from mclstexp_amd import synth
from oracle import ref_cpu

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = ["b8_g785", "b33_g171", "b128_g1000", "b16_g685_vit", "b8_g171_mlp"]
UNTOUCHED_ROW = 60000


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    B, G, D, L, steps = [int(v) for v in z["meta"]]
    return z, dict(B=B, G=G, D=D, layers=L, steps=steps, T=float(z["temperature"]),
                   variant="mlp" if name.endswith("mlp") else "attention")


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

"""Procedural (formula-generated) weights and inputs.

Every value is a pure function of (tensor name, flat index, seed) through a splitmix64
hash, so the 2 x (65536, G) position tables never have to be stored or shipped: tests on
the GPU box, the golden generator in the build container and ``bench.py`` all regenerate
bit-identical fp32 data from the same formula, independent of any torch/numpy RNG.

Shapes/keys follow the reference's ``state_dict``.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def name_seed(name: str, seed: int = 0) -> int:
    return (zlib.crc32(name.encode()) + 0x1000003 * seed) & 0xFFFFFFFF


def uniform01(name: str, idx: np.ndarray, seed: int = 0) -> np.ndarray:
    """u in [0,1) with 24 random bits (exactly representable in fp32)."""
    with np.errstate(over="ignore"):
        key = (idx.astype(np.uint64) + (np.uint64(name_seed(name, seed)) << np.uint64(32))) & _M64
    h = _splitmix64(key)
    return ((h >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def uniform_tensor(name: str, shape: Sequence[int], lo: float, hi: float, seed: int = 0,
                   rows: Optional[np.ndarray] = None) -> torch.Tensor:
    """fp32 tensor of ``shape`` uniform in [lo, hi).  With ``rows`` (for 2-D shapes) only the
    listed rows are generated -> shape (len(rows), shape[1]) -- the same values the full
    tensor holds at those rows."""
    if rows is None:
        n = int(np.prod(shape)) if len(shape) else 1
        idx = np.arange(n, dtype=np.uint64)
        out_shape = tuple(shape)
    else:
        cols = int(shape[1])
        idx = (np.asarray(rows, dtype=np.uint64)[:, None] * np.uint64(cols)
               + np.arange(cols, dtype=np.uint64)[None, :]).reshape(-1)
        out_shape = (len(rows), cols)
    u = uniform01(name, idx, seed)
    v = (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32)
    return torch.from_numpy(v.reshape(out_shape))


SQRT3 = math.sqrt(3.0)


def param_spec(spot_dim: int, image_dim: int, projection_dim: int = 256, heads: int = 8,
               dim_head: int = 64, layers: int = 2, with_tables: bool = True,
               table_rows: int = 65536) -> Dict[str, Tuple[Tuple[int, ...], float, float]]:
    """name -> (shape, lo, hi) for every non-backbone parameter of mclSTExp_Attention
    (/root/reference/model.py:201-223).  Ranges mimic the module defaults' magnitudes:
    Linear U(-1/sqrt(in), 1/sqrt(in)); LayerNorm weight 1 +- 0.05, bias +- 0.05 (perturbed
    so that their gradients are exercised); Embedding unit variance."""
    G, D, P, inner = spot_dim, image_dim, projection_dim, heads * dim_head
    spec: Dict[str, Tuple[Tuple[int, ...], float, float]] = {}
    if with_tables:
        spec["x_embed.weight"] = ((table_rows, G), -SQRT3, SQRT3)
        spec["y_embed.weight"] = ((table_rows, G), -SQRT3, SQRT3)
    for l in range(layers):
        q = f"spot_encoder.{l}."
        kG, kI = 1.0 / math.sqrt(G), 1.0 / math.sqrt(inner)
        spec[q + "attn.norm.weight"] = ((G,), 0.95, 1.05)
        spec[q + "attn.norm.bias"] = ((G,), -0.05, 0.05)
        spec[q + "attn.fn.to_qkv.weight"] = ((3 * inner, G), -kG, kG)
        spec[q + "attn.fn.to_out.0.weight"] = ((G, inner), -kI, kI)
        spec[q + "attn.fn.to_out.0.bias"] = ((G,), -kI, kI)
        spec[q + "ff.norm.weight"] = ((G,), 0.95, 1.05)
        spec[q + "ff.norm.bias"] = ((G,), -0.05, 0.05)
        spec[q + "ff.fn.net.0.weight"] = ((G, G), -kG, kG)
        spec[q + "ff.fn.net.0.bias"] = ((G,), -kG, kG)
        spec[q + "ff.fn.net.3.weight"] = ((G, G), -kG, kG)
        spec[q + "ff.fn.net.3.bias"] = ((G,), -kG, kG)
    for head, d_in in (("image_projection.", D), ("spot_projection.", G)):
        k1, k2 = 1.0 / math.sqrt(d_in), 1.0 / math.sqrt(P)
        spec[head + "projection.weight"] = ((P, d_in), -k1, k1)
        spec[head + "projection.bias"] = ((P,), -k1, k1)
        spec[head + "fc.weight"] = ((P, P), -k2, k2)
        spec[head + "fc.bias"] = ((P,), -k2, k2)
        spec[head + "layer_norm.weight"] = ((P,), 0.95, 1.05)
        spec[head + "layer_norm.bias"] = ((P,), -0.05, 0.05)
    return spec


def make_params(spot_dim: int, image_dim: int, projection_dim: int = 256, heads: int = 8,
                dim_head: int = 64, layers: int = 2, seed: int = 0, with_tables: bool = True,
                table_rows: int = 65536) -> Dict[str, torch.Tensor]:
    spec = param_spec(spot_dim, image_dim, projection_dim, heads, dim_head, layers, with_tables, table_rows)
    return {k: uniform_tensor(k, shp, lo, hi, seed) for k, (shp, lo, hi) in spec.items()}


def make_batch(batch: int, spot_dim: int, image_dim: Optional[int] = None, image_hw: Optional[int] = None,
               seed: int = 0, grid: int = 64, rank: int = 0) -> Dict[str, torch.Tensor]:
    """Synthetic batch in the reference's batch contract (dataset.py:188-195,226-231):
    ``expression`` ~70 % zeros, else log10(1+x)-like values in (0, 4); ``position`` integer-valued
    floats in [0, grid)^2; ``image`` either (B, D) precomputed features (image_dim) or
    (B, 3, H, W) uniform[0,1) pixels (image_hw)."""
    tag = f"@r{rank}"
    u = uniform_tensor("expression" + tag, (batch, spot_dim), 0.0, 1.0, seed)
    expr = torch.where(u < 0.7, torch.zeros_like(u), (u - 0.7) * (4.0 / 0.3))
    pos = torch.floor(uniform_tensor("position" + tag, (batch, 2), 0.0, float(grid), seed))
    out = {"expression": expr.contiguous(), "position": pos.contiguous()}
    if image_dim is not None:
        out["image"] = uniform_tensor("image_features" + tag, (batch, image_dim), -1.0, 1.0, seed)
    elif image_hw is not None:
        out["image"] = uniform_tensor("image" + tag, (batch, 3, image_hw, image_hw), 0.0, 1.0, seed)
    return out


def make_retrieval_case(n_keys: int, n_query: int, dim: int = 256, genes: int = 785, clusters: int = 24,
                        seed: int = 0, duplicates: int = 0) -> Dict[str, np.ndarray]:
    """Synthetic inference-time retrieval problem in the layout the reference's eval scripts hold after their
    transposes (evel_her2st.py:158-172): ``spot_key`` (N, dim) and ``image_query`` (Q, dim) un-normalised
    projection-head outputs (cluster centre + noise, so the cosine top-k is structured like real embeddings),
    ``expression_key`` (N, genes) log-normalised expression (~60 % zeros).  ``duplicates`` > 0 makes the last
    ``duplicates`` keys exact copies of key 0 (ties in the similarity)."""
    cen = uniform_tensor("retr.centres", (clusters, dim), -1.0, 1.0, seed).numpy()
    kc = np.floor(uniform_tensor("retr.key_cluster", (n_keys,), 0.0, float(clusters), seed).numpy()).astype(np.int64)
    qc = np.floor(uniform_tensor("retr.query_cluster", (n_query,), 0.0, float(clusters), seed).numpy()).astype(np.int64)
    key = cen[kc] + 0.8 * uniform_tensor("retr.key_noise", (n_keys, dim), -1.0, 1.0, seed).numpy()
    qry = cen[qc] + 0.8 * uniform_tensor("retr.query_noise", (n_query, dim), -1.0, 1.0, seed).numpy()
    u = uniform_tensor("retr.expression", (n_keys, genes), 0.0, 1.0, seed).numpy()
    expr = np.where(u < 0.6, np.float32(0.0), (u - np.float32(0.6)) * np.float32(10.0)).astype(np.float32)
    key = key.astype(np.float32)
    if duplicates:
        key[n_keys - duplicates:] = key[0]
    return {"spot_key": np.ascontiguousarray(key), "image_query": np.ascontiguousarray(qry.astype(np.float32)),
            "expression_key": np.ascontiguousarray(expr)}

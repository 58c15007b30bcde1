"""CPU restatement (numpy) of the reference's per-spot input preparation.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``; the product path (``mclstexp_amd.input_pipeline``) never
imports this file.

Parity status: patch extraction PINNED -- ``tests/test_oracle_golden.py`` checks these functions against
``tests/golden/input_patches.npz``, produced by ``tests/golden/gen_input_goldens.py`` with PIL (``Image.crop``,
``Image.transpose``, ``Image.rotate``): the library the reference's transform chain bottoms out in (torchvision's
``ToTensor`` / ``TF.hflip`` / ``TF.vflip`` / ``TF.rotate`` on PIL images call exactly these; torchvision itself is an
un-vendored, unpinned dependency that is absent here).  Expression normalisation: "parity unpinned" -- ``scprep`` is an
un-vendored dependency absent here (call site dataset.py:188-189); restated from its published definition
(library_size_normalize: rows rescaled to sum 10^4; transform.log: log10(x + 1)).

Reference citations are relative to /root/reference/.
"""
from __future__ import annotations

import numpy as np


def crop(whole_image: np.ndarray, row: int, col: int, r: int) -> np.ndarray:
    """``img.crop((x-r, y-r, x+r, y+r))`` (dataset.py:226) as an (2r, 2r, 3) uint8 array; outside the image PIL pads
    with zeros."""
    out = np.zeros((2 * r, 2 * r, 3), dtype=np.uint8)
    hs, ws = whole_image.shape[:2]
    r0, c0 = row - r, col - r
    ys, xs = max(r0, 0), max(c0, 0)
    ye, xe = min(r0 + 2 * r, hs), min(c0 + 2 * r, ws)
    if ye > ys and xe > xs:
        out[ys - r0:ye - r0, xs - c0:xe - c0] = whole_image[ys:ye, xs:xe]
    return out


def to_tensor(patch: np.ndarray) -> np.ndarray:
    """transforms.ToTensor (dataset.py:229): uint8 HWC -> float32 CHW / 255."""
    return (patch.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)


def tenx_transform(patch: np.ndarray, hflip: bool, vflip: bool, angle: int) -> np.ndarray:
    """TenxDataset.transform with its random draws made explicit (dataset.py:315-324): TF.hflip, TF.vflip, then
    TF.rotate(angle) -- counter-clockwise, exact for the square patch and multiples of 90 degrees."""
    if hflip:
        patch = patch[:, ::-1]
    if vflip:
        patch = patch[::-1]
    return np.ascontiguousarray(np.rot90(patch, k=(angle % 360) // 90))


def log_library_size_normalize(counts: np.ndarray, rescale: float = 1e4) -> np.ndarray:
    """scp.transform.log(scp.normalize.library_size_normalize(counts)) (dataset.py:188-189)."""
    x = np.asarray(counts, dtype=np.float64)
    s = x.sum(axis=1, keepdims=True)
    f = np.divide(rescale, s, out=np.zeros_like(s), where=s != 0)
    return np.log10(x * f + 1.0)

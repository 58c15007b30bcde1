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


# --------------------------------------------------------------------------- HER2ST / cSCC training augmentation
# dataset.py:63-68: transforms.Compose([ColorJitter(0.5, 0.5, 0.5), RandomHorizontalFlip(), RandomRotation(180),
# ToTensor()]) on the PIL patch.  torchvision (absent here, un-vendored) implements these on PIL images through
# ImageEnhance.Brightness / Contrast / Color (.enhance = Image.blend with the degenerate image), Image.transpose and
# Image.rotate(angle, NEAREST, expand=False, fill 0).  Restated below with the random draws made explicit; PINNED to
# PIL's own outputs (tests/golden/gen_input_goldens.py -> input_augment.npz, tests/test_oracle_golden.py).
def to_luma(rgb: np.ndarray) -> np.ndarray:
    """Image.convert("L") of an RGB uint8 array: ITU-R 601-2 luma in 16.16 fixed point."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(degenerate: np.ndarray, image: np.ndarray, alpha: float) -> np.ndarray:
    """Image.blend(degenerate, image, alpha) on uint8 arrays (libImaging/Blend.c): fp32 arithmetic
    in1 + alpha*(in2 - in1) (multiply and add rounded separately), truncated; clipped when extrapolating."""
    a = np.float32(alpha)
    in1 = degenerate.astype(np.int32)
    d = (image.astype(np.int32) - in1).astype(np.float32)
    t = in1.astype(np.float32) + a * d                      # numpy: two fp32 roundings, like the C expression
    if 0.0 <= float(a) <= 1.0:                              # the C branch tests the float32 argument
        return t.astype(np.int32).astype(np.uint8)          # (UINT8) cast = truncation
    out = np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32)))
    return out.astype(np.uint8)


def adjust_brightness(patch: np.ndarray, f: float) -> np.ndarray:
    return blend(np.zeros_like(patch), patch, f)


def adjust_contrast(patch: np.ndarray, f: float) -> np.ndarray:
    lum = to_luma(patch)
    mean = int(int(lum.astype(np.int64).sum()) / lum.size + 0.5)      # int(ImageStat.Stat(L).mean[0] + 0.5)
    return blend(np.full_like(patch, mean), patch, f)


def adjust_saturation(patch: np.ndarray, f: float) -> np.ndarray:
    lum = to_luma(patch)
    return blend(np.repeat(lum[..., None], 3, axis=2), patch, f)


def color_jitter(patch: np.ndarray, order, brightness: float, contrast: float, saturation: float) -> np.ndarray:
    """ColorJitter.forward with its draws explicit: ``order`` = the permutation of (0 brightness, 1 contrast,
    2 saturation) in which the adjustments are applied (hue is None in the reference's ColorJitter(0.5, 0.5, 0.5))."""
    fns = {0: lambda p: adjust_brightness(p, brightness), 1: lambda p: adjust_contrast(p, contrast),
           2: lambda p: adjust_saturation(p, saturation)}
    for k in order:
        patch = fns[int(k)](patch)
    return patch


def rotation_matrix_fixed(angle: float, w: int, h: int):
    """The six 16.16 fixed-point coefficients libImaging's nearest-neighbour affine path (Geometry.c: affine_fixed)
    derives from Image.rotate's matrix; None for the transpose fast paths (0 / 90 / 180 / 270 on a square image)."""
    import math
    angle = angle % 360.0
    if angle in (0, 180) or (angle in (90, 270) and w == h):
        return None
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + 0.0
    m[2] += cx
    m[5] += cy

    def fix(v):
        v = v * 65536.0 + 0.5
        return int(math.floor(v)) if v < 0.0 else int(v)
    return (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]),
            fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


def rotate_nearest(patch: np.ndarray, angle: float) -> np.ndarray:
    """Image.rotate(angle) (counter-clockwise, NEAREST, expand=False, fill 0) on an (H, W, 3) uint8 array."""
    h, w = patch.shape[:2]
    fx = rotation_matrix_fixed(angle, w, h)
    if fx is None:
        return np.ascontiguousarray(np.rot90(patch, k=int((angle % 360.0) // 90)))
    a0, a1, a2, a3, a4, a5 = fx
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    xin = (a2 + y * a1 + x * a0) >> 16
    yin = (a5 + y * a4 + x * a3) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros_like(patch)
    out[ok] = patch[yin[ok], xin[ok]]
    return out


def her2st_train_transform(patch: np.ndarray, order, brightness: float, contrast: float, saturation: float,
                           hflip: bool, angle: float) -> np.ndarray:
    """dataset.py:63-68 with every random draw explicit -> float32 CHW in [0, 1]."""
    p = color_jitter(patch, order, brightness, contrast, saturation)
    if hflip:
        p = p[:, ::-1]
    return to_tensor(rotate_nearest(np.ascontiguousarray(p), angle))

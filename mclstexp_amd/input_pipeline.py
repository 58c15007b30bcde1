"""Input pipeline on the MI355X: patch extraction from a whole-slide image resident in HBM and the
expression normalisation of the reference's datasets, batched, through the C ABI (csrc/patches.hip).

Reference (paths relative to /root/reference/):
  ViT_HER2ST / SKIN ``__getitem__`` eval branch   dataset.py:226-231  PIL ``crop((x-r, y-r, x+r, y+r))`` + ToTensor
  TenxDataset ``__getitem__`` / ``transform``      dataset.py:315-336  numpy crop of the cv2 (BGR) image, TF.hflip,
                                                                       TF.vflip, TF.rotate(angle in {180, 90, 0, -90})
  expression                                       dataset.py:188-189  scprep log(library_size_normalize(counts))

  ViT_HER2ST / SKIN train branch                   dataset.py:63-68,227  ColorJitter(0.5, 0.5, 0.5), RandomHorizontalFlip,
                                                                       RandomRotation(180), ToTensor -- bit-exact with PIL
No CPU fallback: these functions raise ``RuntimeError`` without a GPU.
"""
from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np
import torch

from . import _lib, ops
from ._lib import check

Tensor = torch.Tensor
ArrayLike = Union[np.ndarray, Tensor]


def _dev() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("mclstexp_amd.input_pipeline: no GPU available (HIP kernels, no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def to_device_image(whole_image: ArrayLike) -> Tensor:
    """(Hs, Ws, 3) uint8 whole-slide image -> contiguous device tensor (call once per slide; it stays in HBM)."""
    t = torch.as_tensor(np.ascontiguousarray(whole_image)) if not isinstance(whole_image, Tensor) else whole_image
    if t.dim() != 3 or t.shape[2] != 3 or t.dtype != torch.uint8:
        raise RuntimeError(f"whole_image: expected (H, W, 3) uint8, got {tuple(t.shape)} {t.dtype}")
    return t.to(_dev()).contiguous()


def extract_patches(whole_image: Tensor, centers_rc: ArrayLike, r: int = 112, hflip: Optional[ArrayLike] = None,
                    vflip: Optional[ArrayLike] = None, rot_k: Optional[ArrayLike] = None, divisor: float = 255.0,
                    layout: str = "nchw_f32") -> Tensor:
    """Batch of (2r x 2r) patches centred at ``centers_rc`` = (N, 2) integer (row, col).  Optional per-patch bool
    ``hflip`` / ``vflip`` and ``rot_k`` (counter-clockwise quarter turns), applied in that order; values are divided by ``divisor`` (255 = ToTensor).  ``layout``:
    "nchw_f32" (the reference's batch tensor) or "nhwc_bf16" (what the backbone kernels consume: a (N,3,2r,2r)
    channels-last bf16 tensor)."""
    img = whole_image if (isinstance(whole_image, Tensor) and whole_image.is_cuda) else to_device_image(whole_image)
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or not img.is_contiguous():
        raise RuntimeError("whole_image: expected a contiguous (H, W, 3) uint8 GPU tensor")
    c = torch.as_tensor(np.asarray(centers_rc) if not isinstance(centers_rc, Tensor) else centers_rc)
    if c.dim() != 2 or c.shape[1] != 2:
        raise RuntimeError(f"centers_rc: expected (N, 2), got {tuple(c.shape)}")
    n = c.shape[0]
    c = c.to(device=img.device, dtype=torch.int32).contiguous()
    opb = None
    if hflip is not None or vflip is not None or rot_k is not None:
        def vec(a, default):
            return torch.full((n,), default, dtype=torch.int64) if a is None else torch.as_tensor(np.asarray(a)).to(torch.int64).cpu()
        code = (vec(hflip, 0) & 1) | ((vec(vflip, 0) & 1) << 1) | ((vec(rot_k, 0) % 4) << 2)
        opb = code.to(torch.uint8).to(img.device)
    p = 2 * r
    out32 = out16 = None
    if layout == "nchw_f32":
        out32 = torch.empty((n, 3, p, p), device=img.device, dtype=torch.float32)
    elif layout == "nhwc_bf16":
        out16 = torch.empty((n, p, p, 3), device=img.device, dtype=torch.bfloat16)
    else:
        raise ValueError("layout must be 'nchw_f32' or 'nhwc_bf16'")
    check(_lib.lib().mcl_patch_gather(img.data_ptr(), img.shape[0], img.shape[1], c.data_ptr(), n, r, ops._p(opb),
                                      float(divisor), ops._p(out32), ops._p(out16), ops._stream()), "mcl_patch_gather")
    return out32 if out32 is not None else out16.permute(0, 3, 1, 2)


def her2st_eval_patches(whole_image: Tensor, centers_xy: ArrayLike, r: int = 112, layout: str = "nchw_f32") -> Tensor:
    """``transforms.ToTensor()(img.crop((x-r, y-r, x+r, y+r)))`` for every spot (dataset.py:226-231, eval branch):
    ``centers_xy`` are the reference's (pixel_x, pixel_y) pairs."""
    c = np.asarray(centers_xy)[:, ::-1]                      # (x, y) -> (row, col)
    return extract_patches(whole_image, np.ascontiguousarray(c), r, divisor=255.0, layout=layout)


def tenx_patches(whole_image_bgr: Tensor, centers_v1v2: ArrayLike, hflip: Sequence[bool], vflip: Sequence[bool],
                 angle: Sequence[int], layout: str = "nchw_f32") -> Tensor:
    """TenxDataset.__getitem__'s image (dataset.py:330-336) for a batch with the random draws given explicitly:
    raw 0-255 values in cv2's channel order, flips, then TF.rotate by ``angle`` in {180, 90, 0, -90}."""
    a = np.asarray(angle)
    if not np.isin(a % 360, (0, 90, 180, 270)).all():
        raise ValueError("TenxDataset.transform only rotates by multiples of 90 degrees")
    return extract_patches(whole_image_bgr, centers_v1v2, 112, hflip, vflip, (a % 360) // 90, divisor=1.0, layout=layout)


def log_library_size_normalize(counts: ArrayLike, rescale: float = 1e4) -> Tensor:
    """scprep.transform.log(scprep.normalize.library_size_normalize(counts)) (dataset.py:188-189) on the GPU:
    (spots, genes) -> log10(counts / row-sum * 1e4 + 1), fp32."""
    x = torch.as_tensor(np.asarray(counts, dtype=np.float32)) if not isinstance(counts, Tensor) else counts
    x = ops._rowmajor(x.to(_dev(), dtype=torch.float32), "counts")
    y = torch.empty_like(x, memory_format=torch.contiguous_format)
    check(_lib.lib().mcl_log_library_size_normalize(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), x.shape[0],
                                                    x.shape[1], float(rescale), ops._stream()),
          "mcl_log_library_size_normalize")
    return y


# --------------------------------------------------------------------------- HER2ST / cSCC training transform
def sample_her2st_draws(n: int, generator: Optional[torch.Generator] = None) -> dict:
    """The random draws of ``transforms.Compose([ColorJitter(0.5, 0.5, 0.5), RandomHorizontalFlip(),
    RandomRotation(degrees=180)])`` (dataset.py:63-68) for ``n`` patches: ``order`` (n, 3) = the order in which the
    three active adjustments run (torchvision draws ``randperm(4)`` over brightness / contrast / saturation / hue;
    hue is None here), factors uniform in [0.5, 1.5], flip with p = 0.5, angle uniform in [-180, 180]."""
    g = generator
    perm = torch.stack([torch.randperm(4, generator=g) for _ in range(n)]) if n else torch.zeros((0, 4), dtype=torch.int64)
    order = torch.stack([row[row != 3] for row in perm]) if n else torch.zeros((0, 3), dtype=torch.int64)
    u = torch.rand((n, 5), generator=g)
    return {"order": order.numpy(), "brightness": (0.5 + u[:, 0]).numpy(), "contrast": (0.5 + u[:, 1]).numpy(),
            "saturation": (0.5 + u[:, 2]).numpy(), "hflip": (u[:, 3] < 0.5).numpy(),
            "angle": (-180.0 + 360.0 * u[:, 4].double()).numpy()}


def _rotation_record(angle: float, p: int):
    """(rot_mode, rot_k, a[6]) for Image.rotate(angle, NEAREST, expand=False) of a p x p patch: quarter-turn fast paths
    as PIL takes them, otherwise libImaging's 16.16 fixed-point affine coefficients (Image.rotate's matrix, rounded to
    15 decimals, then Geometry.c FIX())."""
    import math
    a = float(angle) % 360.0
    if a in (0.0, 90.0, 180.0, 270.0):
        return 0, int(a // 90), [0] * 6
    c = p / 2
    t = -math.radians(a)
    m = [round(math.cos(t), 15), round(math.sin(t), 15), 0.0, round(-math.sin(t), 15), round(math.cos(t), 15), 0.0]
    m[2] = m[0] * -c + m[1] * -c + m[2]
    m[5] = m[3] * -c + m[4] * -c + 0.0
    m[2] += c
    m[5] += c

    def fix(v):
        v = v * 65536.0 + 0.5
        return int(math.floor(v)) if v < 0.0 else int(v)
    return 1, 0, [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]),
                  fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def her2st_train_patches(whole_image: Tensor, centers_xy: ArrayLike, r: int = 112, draws: Optional[dict] = None,
                         generator: Optional[torch.Generator] = None, layout: str = "nchw_f32") -> Tensor:
    """The TRAIN branch of ViT_HER2ST / SKIN ``__getitem__`` (dataset.py:226-228: crop, then self.transforms) for a
    whole batch in one launch: crop around (pixel_x, pixel_y), ColorJitter(0.5, 0.5, 0.5), RandomHorizontalFlip,
    RandomRotation(180), ToTensor.  ``draws`` (see ``sample_her2st_draws``) makes the random choices explicit; omitted,
    they are sampled from ``generator``.  Bit-exact with PIL for given draws (csrc/patches.hip)."""
    img = whole_image if (isinstance(whole_image, Tensor) and whole_image.is_cuda) else to_device_image(whole_image)
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or not img.is_contiguous():
        raise RuntimeError("whole_image: expected a contiguous (H, W, 3) uint8 GPU tensor")
    cxy = np.asarray(centers_xy)
    if cxy.ndim != 2 or cxy.shape[1] != 2:
        raise RuntimeError(f"centers_xy: expected (N, 2), got {cxy.shape}")
    n = cxy.shape[0]
    if draws is None:
        draws = sample_her2st_draws(n, generator)
    p = 2 * r
    rec = np.zeros((n, 16), dtype=np.int32)
    fac = np.zeros((n, 3), dtype=np.float32)
    for i in range(n):
        o = [int(v) for v in draws["order"][i]]
        if sorted(o) != [0, 1, 2]:
            raise ValueError("draws['order'] rows must be permutations of (0, 1, 2)")
        mode, k, a = _rotation_record(float(draws["angle"][i]), p)
        rec[i, 0] = o[0] | (o[1] << 2) | (o[2] << 4)
        rec[i, 1] = int(bool(draws["hflip"][i]))
        rec[i, 2], rec[i, 3] = mode, k
        rec[i, 7:13] = a
        fac[i] = (draws["brightness"][i], draws["contrast"][i], draws["saturation"][i])
    rec[:, 4:7] = fac.view(np.int32)
    prm = torch.from_numpy(rec).to(img.device)
    c = torch.from_numpy(np.ascontiguousarray(cxy[:, ::-1]).astype(np.int32)).to(img.device)      # (x, y) -> (row, col)
    out32 = out16 = None
    if layout == "nchw_f32":
        out32 = torch.empty((n, 3, p, p), device=img.device, dtype=torch.float32)
    elif layout == "nhwc_bf16":
        out16 = torch.empty((n, p, p, 3), device=img.device, dtype=torch.bfloat16)
    else:
        raise ValueError("layout must be 'nchw_f32' or 'nhwc_bf16'")
    check(_lib.lib().mcl_her2st_train_patches(img.data_ptr(), img.shape[0], img.shape[1], c.data_ptr(), n, r,
                                              prm.data_ptr(), 255.0, ops._p(out32), ops._p(out16), ops._stream()),
          "mcl_her2st_train_patches")
    return out32 if out32 is not None else out16.permute(0, 3, 1, 2)

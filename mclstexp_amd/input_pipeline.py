"""Input pipeline on the MI355X: patch extraction from a whole-slide image resident in HBM and the
expression normalisation of the reference's datasets, batched, through the C ABI (csrc/patches.hip).

Reference (paths relative to /root/reference/):
  ViT_HER2ST / SKIN ``__getitem__`` eval branch   dataset.py:226-231  PIL ``crop((x-r, y-r, x+r, y+r))`` + ToTensor
  TenxDataset ``__getitem__`` / ``transform``      dataset.py:315-336  numpy crop of the cv2 (BGR) image, TF.hflip,
                                                                       TF.vflip, TF.rotate(angle in {180, 90, 0, -90})
  expression                                       dataset.py:188-189  scprep log(library_size_normalize(counts))

Not built: the train-time ColorJitter / arbitrary-angle rotation of the HER2ST / cSCC datasets (dataset.py:63-68).
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

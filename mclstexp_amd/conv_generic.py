"""Generic convolution / pooling on this library's kernels for BOTH activation types (fp32 and bf16), channels-last.

Used where the specialised DenseNet kernels (csrc/dense_conv.hip ...) do not apply:
  * fp32 activations -- ``backbone_dtype=None``, the reference-numerics mode of the DenseNet backbone
    (/root/reference/model.py:72-85 is pure fp32): no MIOpen / ATen convolution or pooling call remains;
  * the ResNet encoders (/root/reference/model.py:88-148; resnet_fused.py).

A convolution is lowered to ``im2col`` (csrc/im2col.hip) + a GEMM of this library: ``mcl_gemm`` (exact fp32 MFMA,
v_mfma_f32_32x32x2_f32) for fp32 activations, ``mcl_gemm_bf16`` for bf16.  The column order of im2col is the storage order of
a channels-last weight (C_out, kh, kw, C_in), so the weight is consumed in place; a 1x1 stride-1 convolution skips im2col (the
NHWC activation IS the GEMM operand, channel slices of a wider buffer included).  Backward-data = GEMM + ``col2im`` (a
deterministic gather), weight gradient = split-K GEMM with a fixed-order merge, accumulated straight into a dense fp32
``.grad`` when the parameter owns one.  Forward and backward are hand-written ``torch.autograd.Function``s, as everywhere in
this package (the reference gets its backward from autograd, train.py:38).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib, ops
from ._lib import check

Tensor = torch.Tensor
CL = torch.channels_last
A_KM, B_KM, OUT_F32 = 1, 2, 16           # mcl_gemm_bf16 flags (include/mclstexp_hip.h)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dt(t: Tensor) -> int:
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise RuntimeError(f"conv_generic: unsupported activation dtype {t.dtype}")


def _rows(t: Tensor) -> Tuple[int, int, int, int]:
    from .densenet_fused import _rows as r
    return r(t)


def _dense_cl(t: Tensor) -> Tensor:
    from .densenet_fused import dense_cl
    return dense_cl(t)


def _out_hw(H: int, W: int, k: int, stride: int, pad: int) -> Tuple[int, int]:
    return (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1


def im2col(x: Tensor, k: int, stride: int, pad: int) -> Tensor:
    """(B, C, H, W) channels-last (channel slices allowed) -> (B*OH*OW, k*k*C) contiguous, column order (ky, kx, c)."""
    B, C, H, W = x.shape
    px, S, _, ld = _rows(x)
    OH, OW = _out_hw(H, W, k, stride, pad)
    cols = torch.empty((B * OH * OW, k * k * C), device=x.device, dtype=x.dtype)
    check(_lib.lib().mcl_im2col_nhwc(px, ld, B, H, W, C, k, k, stride, pad, _dt(x), cols.data_ptr(), _stream()),
          "mcl_im2col_nhwc")
    return cols


def col2im(dcols: Tensor, shape, k: int, stride: int, pad: int, out: Optional[Tensor] = None, accumulate: bool = False
           ) -> Tensor:
    B, C, H, W = shape
    if out is None:
        out = torch.empty((B, C, H, W), device=dcols.device, dtype=dcols.dtype, memory_format=CL)
    po, S, Co, ld = _rows(out)
    assert Co == C and out.dtype == dcols.dtype
    check(_lib.lib().mcl_col2im_nhwc(dcols.data_ptr(), B, H, W, C, k, k, stride, pad, _dt(dcols), po, ld, int(accumulate),
                                     _stream()), "mcl_col2im_nhwc")
    return out


def _gemm(A: Tensor, a_ptr: int, lda: int, a_km: bool, B: Tensor, b_ptr: int, ldb: int, b_km: bool, C: Tensor, c_ptr: int,
          ldc: int, M: int, N: int, K: int, out_f32: bool = False, accumulate: bool = False, ksplit: int = 1) -> None:
    """C (M x N) (+)= A (M x K) . B (K x N) on the GEMM of the operands' dtype.  ``a_km``: A stored [K][M] (reduction-major),
    else [M][K]; ``b_km``: B stored [K][N], else [N][K]."""
    L = _lib.lib()
    if A.dtype == torch.bfloat16:
        flags = (A_KM if a_km else 0) | (B_KM if b_km else 0) | (OUT_F32 if out_f32 else 0)
        ws = None
        if ksplit > 1:
            from .densenet_fused import _ws
            ws = _ws(L.mcl_gemm_bf16_workspace_floats(M, ldc, ksplit), C.device)
        check(L.mcl_gemm_bf16(a_ptr, lda, 0, b_ptr, ldb, 0, c_ptr, ldc, 0, M, N, K, 1, 1, 0, 0, 0, 1.0, flags, None, None, 0, 0,
                              None, 0, None, 0, ksplit, None if ws is None else ws.data_ptr(), int(accumulate), _stream()),
              "mcl_gemm_bf16")
        return
    # fp32: mcl_gemm with explicit element strides (exact fp32 MFMA); split-K is chosen inside gemm_raw
    sAm, sAk = (1, lda) if a_km else (lda, 1)
    sBk, sBn = (ldb, 1) if b_km else (1, ldb)
    a_off = (a_ptr - A.data_ptr()) // 4
    b_off = (b_ptr - B.data_ptr()) // 4
    c_off = (c_ptr - C.data_ptr()) // 4
    ops.gemm_raw(M, N, K, 1, A, sAm, sAk, 0, B, sBk, sBn, 0, C, ldc, 0, flags=ops.EPI_ACCUM if accumulate else 0,
                 a_off=a_off, b_off=b_off, c_off=c_off, compute=ops.COMPUTE_F32)


def _weight_khwc(w: Tensor, dt: torch.dtype) -> Tensor:
    """The weight in the activations' dtype with (C_out, kh, kw, C_in) storage (a channels-last parameter is that already;
    FusedAdam's bf16 shadow is used when attached)."""
    from .densenet_fused import _weight
    return _weight(w, dt)


# The unfolded patches of a k > 1 convolution are kept from the forward for the weight gradient (one im2col launch per convolution
# and step instead of two: 17 % of the ResNet-18 step, 8 % of the fp32 DenseNet step) when they fit this many bytes -- HBM is
# 288 GB: the fp32 DenseNet's 58 unfolded 3 x 3 inputs are 20 GB at batch 128.
KEEP_COLS_MAX_BYTES = 3 << 30


def conv_fwd(x: Tensor, wk: Tensor, stride: int, pad: int, out: Optional[Tensor] = None, want_cols: bool = False):
    """y = conv2d(x, w, stride, pad) (no bias).  x (B, Ci, H, W) channels-last view, wk (Co, Ci, k, k) channels-last storage in
    x's dtype; ``out``: optional channels-last (sliced) destination.  ``want_cols``: returns (y, unfolded patches or None)."""
    B, Ci, H, W = x.shape
    Co, _, k, _ = wk.shape
    OH, OW = _out_hw(H, W, k, stride, pad)
    y = out if out is not None else torch.empty((B, Co, OH, OW), device=x.device, dtype=x.dtype, memory_format=CL)
    py, Sy, Cy, ldy = _rows(y)
    assert (Sy, Cy) == (B * OH * OW, Co) and y.dtype == x.dtype
    if k == 1 and stride == 1 and pad == 0:
        pa, S, _, lda = _rows(x)
        A = x
    else:
        A = im2col(x, k, stride, pad)
        pa, lda = A.data_ptr(), k * k * Ci
    _gemm(A, pa, lda, False, wk, wk.data_ptr(), k * k * Ci, False, y, py, ldy, B * OH * OW, Co, k * k * Ci)
    if want_cols:
        keep = A is not x and A.numel() * A.element_size() <= KEEP_COLS_MAX_BYTES
        return y, (A if keep else None)
    return y


def conv_bwd_data(dy: Tensor, wk: Tensor, x_shape, stride: int, pad: int) -> Tensor:
    B, Ci, H, W = x_shape
    Co, _, k, _ = wk.shape
    pd, S, Cd, ldd = _rows(dy)
    assert Cd == Co
    K = k * k * Ci
    if k == 1 and stride == 1 and pad == 0:
        dx = torch.empty((B, Ci, H, W), device=dy.device, dtype=dy.dtype, memory_format=CL)
        _gemm(dy, pd, ldd, False, wk, wk.data_ptr(), K, True, dx, dx.data_ptr(), Ci, S, K, Co)
        return dx
    if stride == 1 and 2 * pad == k - 1 and 2 * Co <= Ci:
        # a "same" convolution with fewer outputs than inputs (DenseNet's 3 x 3: 128 -> 32): the data gradient is itself a same
        # convolution of dy with the kernel rotated by 180 degrees and its channel roles swapped.  Unfolding dy costs k*k*Co columns
        # per pixel instead of the k*k*Ci of the column-gradient buffer (a quarter here), and there is no col2im pass at all
        # (fp32 DenseNet step: 11.1 + 4.7 ms of column-gradient GEMMs and col2im).
        dyc = im2col(dy, k, 1, pad)                                       # (S, k*k*Co), columns (ky, kx, co)
        # (Ci, k, k, Co): wf[ci,ky,kx,co] = w[co,k-1-ky,k-1-kx,ci] -- one own launch (was permute + flip + contiguous on ATen)
        wf = torch.empty((Ci, k * k * Co), device=dy.device, dtype=dy.dtype)
        check(_lib.lib().mcl_weight_rot180(wk.data_ptr(), wf.data_ptr(), Co, k, Ci, _dt(dy), _stream()), "mcl_weight_rot180")
        dx = torch.empty((B, Ci, H, W), device=dy.device, dtype=dy.dtype, memory_format=CL)
        _gemm(dyc, dyc.data_ptr(), k * k * Co, False, wf, wf.data_ptr(), k * k * Co, False, dx, dx.data_ptr(), Ci, S, Ci, k * k * Co)
        return dx
    dcols = torch.empty((S, K), device=dy.device, dtype=dy.dtype)
    _gemm(dy, pd, ldd, False, wk, wk.data_ptr(), K, True, dcols, dcols.data_ptr(), K, S, K, Co)
    return col2im(dcols, (B, Ci, H, W), k, stride, pad)


def conv_bwd_weight(dy: Tensor, x: Tensor, w_param: Tensor, stride: int, pad: int, cols: Optional[Tensor] = None
                    ) -> Optional[Tensor]:
    """dW (Co, kh, kw, Ci) = dy^T . cols.  Added straight into ``w_param.grad`` when it is a dense fp32 tensor of
    (Co, kh, kw, Ci) storage (returns None), else returned as a fresh fp32 tensor shaped like the parameter.  ``cols``: the
    forward's unfolded patches, when it kept them."""
    from .densenet_fused import DIRECT_PARAM_GRADS, _direct_grad_ok
    Co, Ci, k, _ = w_param.shape
    pd, S, Cd, ldd = _rows(dy)
    assert Cd == Co
    K = k * k * Ci
    if k == 1 and stride == 1 and pad == 0:
        pa, _, _, lda = _rows(x)
        A = x
    else:
        A = cols if cols is not None else im2col(x, k, stride, pad)
        pa, lda = A.data_ptr(), K
    direct = (DIRECT_PARAM_GRADS and _direct_grad_ok(w_param) and w_param.grad.permute(0, 2, 3, 1).is_contiguous())
    if direct:
        tgt = w_param.grad
    else:
        tgt = torch.zeros((Co, Ci, k, k), device=dy.device, dtype=torch.float32).contiguous(memory_format=CL)
    if dy.dtype == torch.bfloat16:
        tiles = ((Co + 255) // 256) * ((K + 255) // 256)
        ksplit = max(2, min(64, (256 + tiles - 1) // tiles))
        _gemm(dy, pd, ldd, True, A, pa, lda, True, tgt, tgt.data_ptr(), K, Co, K, S, out_f32=True, accumulate=True,
              ksplit=ksplit)
    else:
        _gemm(dy, pd, ldd, True, A, pa, lda, True, tgt, tgt.data_ptr(), K, Co, K, S, accumulate=True)
    return None if direct else tgt


# Test instrumentation (tests/test_resnet_layerwise_gpu.py): when a list, every ConvFn records the operands its kernels read
# (x, the cast weight) and what they produced (y; in the backward dy, dx and -- when the weight gradient went straight into
# .grad -- the parameter), so that each convolution can be re-evaluated in fp64 from exactly those operands.
CAPTURE_CONVS: Optional[list] = None


class ConvFn(torch.autograd.Function):
    """conv2d(x, w, stride, padding) without bias on channels-last activations (fp32 or bf16)."""

    @staticmethod
    def forward(ctx, x, w, stride, pad):
        wk = _weight_khwc(w, x.dtype)
        y, cols = conv_fwd(x, wk, stride, pad, want_cols=True)
        if not ctx.needs_input_grad[1]:
            cols = None
        ctx.has_cols = cols is not None
        if cols is not None:
            ctx.save_for_backward(x, wk, cols)
        else:
            ctx.save_for_backward(x, wk)
        ctx.w, ctx.geo = w, (stride, pad)
        ctx.cap = None
        if CAPTURE_CONVS is not None:
            ctx.cap = {"x": x, "wk": wk, "y": y, "stride": stride, "pad": pad, "param": w}
            CAPTURE_CONVS.append(ctx.cap)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wk = ctx.saved_tensors[:2]
        cols = ctx.saved_tensors[2] if ctx.has_cols else None
        stride, pad = ctx.geo
        if not dy.is_contiguous(memory_format=CL):
            dy = _dense_cl(dy)
        g_before = None
        if ctx.cap is not None and getattr(ctx.w, "grad", None) is not None:
            g_before = ctx.w.grad.detach().clone()
        dw = conv_bwd_weight(dy, x, ctx.w, stride, pad, cols) if ctx.needs_input_grad[1] else None
        dx = conv_bwd_data(dy, wk, x.shape, stride, pad) if ctx.needs_input_grad[0] else None
        if ctx.cap is not None:
            ctx.cap.update({"dy": dy, "dx": dx})
            if dw is not None:
                ctx.cap["dw"] = dw.detach().clone()
            elif getattr(ctx.w, "grad", None) is not None:
                ctx.cap["dw"] = ctx.w.grad.detach() - g_before if g_before is not None else ctx.w.grad.detach().clone()
        return dx, (None if dw is None else dw.to(ctx.w.dtype)), None, None


def conv2d(x: Tensor, w: Tensor, stride: int = 1, padding: int = 0) -> Tensor:
    return ConvFn.apply(x, w, stride, padding)


# --------------------------------------------------------------------------------------------------------------- pooling
class MaxPool3s2Fn(torch.autograd.Function):
    """nn.MaxPool2d(3, stride=2, padding=1), fp32 or bf16 channels-last."""

    @staticmethod
    def forward(ctx, x):
        x = _dense_cl(x)
        B, C, H, W = x.shape
        y = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=x.device, dtype=x.dtype, memory_format=CL)
        idx = torch.empty((B, y.shape[2], y.shape[3], C), device=x.device, dtype=torch.uint8)
        check(_lib.lib().mcl_maxpool3s2_nhwc_fwd_any(x.data_ptr(), y.data_ptr(), idx.data_ptr(), B, H, W, C, _dt(x), _stream()),
              "mcl_maxpool3s2_nhwc_fwd_any")
        ctx.save_for_backward(idx)
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, C, H, W = ctx.shape
        dy = _dense_cl(dy)
        dx = torch.empty((B, C, H, W), device=dy.device, dtype=dy.dtype, memory_format=CL)
        check(_lib.lib().mcl_maxpool3s2_nhwc_bwd_any(idx.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, H, W, C, _dt(dy),
                                                     _stream()), "mcl_maxpool3s2_nhwc_bwd_any")
        return dx


class AvgPool2Fn(torch.autograd.Function):
    """nn.AvgPool2d(2, 2) (floor on odd maps), fp32 or bf16 channels-last."""

    @staticmethod
    def forward(ctx, x):
        x = _dense_cl(x)
        B, C, H, W = x.shape
        y = torch.empty((B, C, H // 2, W // 2), device=x.device, dtype=x.dtype, memory_format=CL)
        check(_lib.lib().mcl_avgpool2_nhwc_any(x.data_ptr(), y.data_ptr(), B, H, W, C, 0, _dt(x), _stream()),
              "mcl_avgpool2_nhwc_any")
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = _dense_cl(dy)
        dx = torch.empty((B, C, H, W), device=dy.device, dtype=dy.dtype, memory_format=CL)
        check(_lib.lib().mcl_avgpool2_nhwc_any(dy.data_ptr(), dx.data_ptr(), B, H, W, C, 1, _dt(dy), _stream()),
              "mcl_avgpool2_nhwc_any")
        return dx


class GlobalAvgPoolFn(torch.autograd.Function):
    """F.adaptive_avg_pool2d(x, (1, 1)).flatten(1) -> (B, C) fp32 (model.py:83-84, 98-99)."""

    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        px, S, _, ld = _rows(x)
        out = torch.empty((B, C), device=x.device, dtype=torch.float32)
        check(_lib.lib().mcl_gap_nhwc_fwd(px, ld, B, H * W, C, _dt(x), out.data_ptr(), _stream()), "mcl_gap_nhwc_fwd")
        ctx.shape, ctx.dt = (B, C, H, W), x.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        g = g.contiguous().float()
        dx = torch.empty((B, C, H, W), device=g.device, dtype=ctx.dt, memory_format=CL)
        check(_lib.lib().mcl_gap_nhwc_bwd(g.data_ptr(), B, H * W, C, 0 if ctx.dt == torch.float32 else 1, dx.data_ptr(),
                                          _stream()), "mcl_gap_nhwc_bwd")
        return dx


class AddReluFn(torch.autograd.Function):
    """relu(a + b): the residual join of a ResNet block (model.py:88-148 via torchvision's BasicBlock / Bottleneck)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _dense_cl(a), _dense_cl(b)
        y = torch.empty_like(a, memory_format=CL)
        check(_lib.lib().mcl_add_relu(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), 0, _dt(a), _stream()), "mcl_add_relu")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _dense_cl(dy)
        dx = torch.empty_like(y, memory_format=CL)
        check(_lib.lib().mcl_add_relu(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel(), 1, _dt(y), _stream()),
              "mcl_add_relu (backward)")
        return dx, dx


class Fork2Fn(torch.autograd.Function):
    """x -> (x, x) for a tensor with two consumers (the input of a residual block feeds the block and its shortcut): the
    backward adds the two gradients with this library's kernel instead of autograd's ATen accumulation."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        g1, g2 = _dense_cl(g1), _dense_cl(g2)
        out = torch.empty_like(g1, memory_format=CL)
        check(_lib.lib().mcl_add_relu(g1.data_ptr(), g2.data_ptr(), out.data_ptr(), g1.numel(), 2, _dt(g1), _stream()),
              "mcl_add_relu (plain add)")
        return out


def fork2(x: Tensor):
    return Fork2Fn.apply(x) if (x.requires_grad and torch.is_grad_enabled()) else (x, x)


def max_pool_3s2(x: Tensor) -> Tensor:
    return MaxPool3s2Fn.apply(x)


def avg_pool_2(x: Tensor) -> Tensor:
    return AvgPool2Fn.apply(x)


def global_avg_pool(x: Tensor) -> Tensor:
    return GlobalAvgPoolFn.apply(x)


def add_relu(a: Tensor, b: Tensor) -> Tensor:
    return AddReluFn.apply(a, b)

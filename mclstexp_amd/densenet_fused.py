"""Concat-free, stats-caching execution of the DenseNet-121 feature extractor.

Same module tree and parameters as ``backbones.densenet121_features_module`` (torchvision layout, what
/root/reference/model.py:75-76 wraps) -- only the execution differs.  Round-1 profiling showed the
stock path spends ~60 % of the training step in HBM-bound BatchNorm / ReLU / torch.cat / gradient-add
kernels.  Here:

  * each dense block owns ONE channels-last buffer (B, H, W, C_total); a layer reads its input as the
    channel slice [:C_in] in place and its 32 new channels are written into [C_in:C_in+32] -- there is
    no torch.cat and no O(L^2) copy;
  * BatchNorm batch statistics of a feature map are computed once, when the map is produced (fused with
    the copy into the buffer); every later norm1 / transition norm / norm5 that consumes those channels
    re-uses them (they are the same numbers torch would recompute per layer);
  * BN+ReLU forward is one read + one write (csrc/bnrelu.hip); BN+ReLU backward is a reduce pass and a
    dx pass that ACCUMULATES in place into the block's gradient buffer, replacing autograd's per-layer
    slice gradients and their add chain;
  * in bf16 every convolution, its data gradient and its weight gradient run on the hand-written kernels of
    csrc/dense_conv.hip, conv3x3_rows.hip, dense_bwd.hip and wrw_fused.hip; with fp32 activations (``backbone_dtype=None``,
    the reference-numerics mode) and for any shape the specialised kernels do not cover, a convolution is im2col + this
    library's own GEMM (conv_generic.py: exact fp32 MFMA for fp32 activations) between the same BatchNorm kernels, and the
    pools are csrc/pool_generic.hip -- no MIOpen / ATen convolution or pooling call exists in this file (round 4).

Train-mode semantics of nn.BatchNorm2d are kept: batch statistics, running_mean / running_var (unbiased)
/ num_batches_tracked updates with momentum 0.1 (batched with torch._foreach ops at the end of forward).
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from ._lib import check

Tensor = torch.Tensor
CL = torch.channels_last
_ws_cache = {}

# Round 1-3 counted here every place where this file left the hand-written kernels for a library path (MIOpen through
# aten.convolution*, ATen pooling).  No such place remains: shapes outside the specialised kernels run on the generic
# im2col + own-GEMM path (conv_generic.py).  The counter API stays for its callers (tests, bench.py's JSON line): always empty.
_fallbacks: dict = {}


def fallback_counts() -> dict:
    return dict(_fallbacks)


def reset_fallbacks() -> None:
    _fallbacks.clear()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---- MCL_STAMPS=1: GPU wall-clock stamps at labelled points of the step (a one-thread kernel on the CURRENT stream, captured
# into the step graph like any other launch): the untraced timeline of a replayed step (tools/step_stamps.py).
STAMPS = os.environ.get("MCL_STAMPS", "0") == "1"
_stamp_labels: list = []
_stamp_buf = {}


def stamp(label: str) -> None:
    if not STAMPS:
        return
    dev = torch.cuda.current_device()
    buf = _stamp_buf.get(dev)
    if buf is None:
        buf = torch.zeros(4096, device=f"cuda:{dev}", dtype=torch.int64)
        _stamp_buf[dev] = buf
    if label in _stamp_labels:
        idx = _stamp_labels.index(label)
    else:
        _stamp_labels.append(label)
        idx = len(_stamp_labels) - 1
    check(_lib.lib().mcl_stamp(buf.data_ptr(), idx, _stream()), "mcl_stamp")


def read_stamps() -> dict:
    """{label: GPU wall clock in microseconds} of the last execution of every stamp."""
    torch.cuda.synchronize()
    buf = _stamp_buf.get(torch.cuda.current_device())
    if buf is None:
        return {}
    v = buf[:len(_stamp_labels)].cpu().tolist()
    return {lab: t / 100.0 for lab, t in zip(_stamp_labels, v)}


def _dt(t: Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return 1
    if t.dtype == torch.float32:
        return 0
    raise RuntimeError(f"densenet_fused: unsupported activation dtype {t.dtype}")


def _rows(t: Tensor) -> Tuple[int, int, int, int]:
    """(ptr, S, C, ld) of a channels-last (possibly channel-sliced) 4-D tensor."""
    if not t.is_cuda:
        raise RuntimeError("densenet_fused: tensors must be on the GPU (no CPU fallback)")
    B, C, H, W = t.shape
    ld = t.stride(3) if W > 1 else (t.stride(2) if H > 1 else t.stride(0))
    ok = t.stride(1) == 1 and (W == 1 or t.stride(3) == ld) and (H == 1 or t.stride(2) == W * ld) and \
        (B == 1 or t.stride(0) == H * W * ld)
    if not ok:
        raise RuntimeError(f"densenet_fused: expected a channels-last view, got shape {tuple(t.shape)} "
                           f"strides {t.stride()}")
    return t.data_ptr(), B * H * W, C, ld


def dense_cl(t: Tensor) -> Tensor:
    """``t.contiguous(memory_format=channels_last)`` on an own kernel for the case that occurs inside the encoders: ``t`` is a
    CHANNEL SLICE of a wider channels-last buffer (rows of C elements at a larger row stride) -- one 2-D copy (mcl_copy_rows)
    instead of an ATen strided copy.  A tensor that is dense channels-last already is returned as is; a genuinely different
    layout (an NCHW batch) keeps the torch conversion."""
    if t.is_contiguous(memory_format=CL):
        return t
    B, C, H, W = t.shape
    ld = t.stride(3) if W > 1 else (t.stride(2) if H > 1 else t.stride(0))
    ok = (t.is_cuda and t.stride(1) == 1 and (W == 1 or t.stride(3) == ld) and (H == 1 or t.stride(2) == W * ld)
          and (B == 1 or t.stride(0) == H * W * ld) and ld >= C)
    if not ok:
        return t.contiguous(memory_format=CL)
    out = torch.empty((B, C, H, W), device=t.device, dtype=t.dtype, memory_format=CL)
    es = t.element_size()
    check(_lib.lib().mcl_copy_rows(t.data_ptr(), ld * es, out.data_ptr(), C * es, B * H * W, C * es, _stream()), "mcl_copy_rows")
    return out


def _ws(nfloats: int, device) -> Tensor:
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    w = _ws_cache.get(key)
    if w is None or w.numel() < nfloats:
        w = torch.empty(max(nfloats, 1 << 20), device=device, dtype=torch.float32)
        _ws_cache[key] = w
    return w


def bn_stats(x: Tensor, mean: Tensor, var: Tensor, rstd: Tensor, eps: float, copy_out: Optional[Tensor] = None):
    p, S, C, ld = _rows(x)
    dt = _dt(x)
    L = _lib.lib()
    n = L.mcl_bn_workspace_floats(S, C, dt)
    if n < 0:
        raise RuntimeError(f"mcl_bn_workspace_floats rejected S={S} C={C} dtype={dt}")
    ws = _ws(n, x.device)
    po, ldo = (None, 0)
    if copy_out is not None:
        po, S2, C2, ldo = _rows(copy_out)
        assert (S2, C2) == (S, C) and copy_out.dtype == x.dtype
    check(L.mcl_bn_stats(p, ld, S, C, dt, po, ldo, ws.data_ptr(), eps, mean.data_ptr(), var.data_ptr(),
                         rstd.data_ptr(), _stream()), "mcl_bn_stats")


def bn_act_fwd(x: Tensor, gamma: Tensor, beta: Tensor, mean: Tensor, rstd: Tensor, relu: bool, out: Tensor):
    p, S, C, ld = _rows(x)
    po, S2, C2, ldo = _rows(out)
    assert (S2, C2) == (S, C) and out.dtype == x.dtype
    check(_lib.lib().mcl_bn_act_fwd(p, ld, S, C, _dt(x), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                    rstd.data_ptr(), int(relu), po, ldo, _stream()), "mcl_bn_act_fwd")


def _has_grad_hooks(p: Tensor) -> bool:
    """Tensor hooks / post-accumulate-grad hooks registered on a parameter: they fire from autograd's AccumulateGrad node, which
    a gradient added straight into ``.grad`` by a kernel never reaches -- such a parameter takes the autograd hand-over."""
    return bool(getattr(p, "_backward_hooks", None)) or bool(getattr(p, "_post_accumulate_grad_hooks", None))


def _direct_grad_possible(p: Tensor) -> bool:
    """PURE predicate: may the backward kernels add this parameter's gradient straight into ``p.grad``?  True when it owns a
    dense fp32 .grad of its own layout (FusedAdam's flat bucket) -- or when it could be given one (``_ensure_dense_grad``)."""
    if _has_grad_hooks(p):
        return False
    g = getattr(p, "grad", None)
    if g is None:
        return (DIRECT_PARAM_GRADS and isinstance(p, torch.nn.Parameter) and p.requires_grad and p.is_cuda
                and p.dtype == torch.float32)
    return (g.dtype == torch.float32 and g.is_cuda and g.shape == p.shape and g.stride() == p.stride()
            and not g.requires_grad)


def _ensure_dense_grad(p: Tensor) -> None:
    """A parameter whose .grad is None -- the state torch.optim.*.zero_grad() (set_to_none) leaves behind,
    /root/reference/train.py:37 -- gets a zero-filled dense fp32 .grad, so the reference's own ``Adam`` + ``zero_grad()`` +
    ``backward()`` loop stays on the HIP weight-gradient kernels (they accumulate; nothing is handed back to autograd)."""
    if getattr(p, "grad", None) is None:
        p.grad = torch.zeros_like(p)                  # preserve_format: same strides as the parameter


def _direct_grad_ok(p: Tensor) -> bool:
    """``_direct_grad_possible`` + ``_ensure_dense_grad``: called where a kernel that accumulates into ``p.grad`` is about to
    be launched.  Consequences of the direct path (documented): the backward returns None for such a parameter,
    so ``torch.autograd.grad(loss, params)`` is NOT supported for them (use ``.backward()`` and read ``.grad``), and a
    parameter with tensor / post-accumulate-grad hooks is excluded (its gradient goes through autograd as usual)."""
    if not _direct_grad_possible(p):
        return False
    _ensure_dense_grad(p)
    return True


def bn_act_bwd(dy: Tensor, x: Tensor, gamma: Tensor, beta: Tensor, mean: Tensor, rstd: Tensor, relu: bool,
               dx: Tensor, accumulate: bool, into_param_grads: bool = False
               ) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """``into_param_grads``: add dgamma/dbeta straight into ``gamma.grad`` / ``beta.grad`` (they must exist:
    the flat FusedAdam bucket) and return (None, None) -- no temporaries, no AccumulateGrad add kernels."""
    p, S, C, ld = _rows(x)
    pd, S2, C2, ldd = _rows(dy)
    px, S3, C3, ldx = _rows(dx)
    assert (S2, C2) == (S, C) == (S3, C3) and dy.dtype == x.dtype == dx.dtype
    dt = _dt(x)
    L = _lib.lib()
    ws = _ws(L.mcl_bn_workspace_floats(S, C, dt), x.device)
    if into_param_grads:
        dg, db = gamma.grad, beta.grad
    else:
        dg = torch.empty(C, device=x.device, dtype=torch.float32)
        db = torch.empty(C, device=x.device, dtype=torch.float32)
    check(L.mcl_bn_act_bwd(pd, ldd, p, ld, S, C, dt, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                           rstd.data_ptr(), int(relu), ws.data_ptr(), dg.data_ptr(), db.data_ptr(),
                           int(into_param_grads), px, ldx, int(accumulate), _stream()), "mcl_bn_act_bwd")
    return (None, None) if into_param_grads else (dg, db)


# When a parameter already owns a dense fp32 .grad (FusedAdam's flat bucket, zeroed every step), backward
# adds its gradient straight into it and returns None to autograd: saves one temporary, one dtype cast and
# one AccumulateGrad add kernel per parameter (~480 tiny launches per DenseNet-121 step).
DIRECT_PARAM_GRADS = True          # (module constant: tests / tools may flip it; the environment switch is gone)


class BNActFn(torch.autograd.Function):
    """y = relu?(BatchNorm_train(x)) with the batch statistics supplied (they are functions of x: the
    backward is the full train-mode BatchNorm backward)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, rstd, relu):
        y = torch.empty_like(x, memory_format=CL)
        bn_act_fwd(x, gamma, beta, mean, rstd, relu, y)
        ctx.save_for_backward(x, mean, rstd)
        ctx.relu = relu
        ctx.params = (gamma, beta)      # the Parameter objects themselves (their .grad may be written directly)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        dy = dense_cl(dy)
        dx = torch.empty_like(x, memory_format=CL)
        direct = DIRECT_PARAM_GRADS and _direct_grad_ok(gamma) and _direct_grad_ok(beta)
        dg, db = bn_act_bwd(dy, x, gamma, beta, mean, rstd, ctx.relu, dx, False, into_param_grads=direct)
        return dx, dg, db, None, None, None


class _RunningStats:
    """Collects (module, mean, biased var, count) during forward; applies nn.BatchNorm2d's running-stat
    updates with a few multi-tensor launches at the end."""

    def __init__(self):
        self.mods: List[nn.BatchNorm2d] = []
        self.means: List[Tensor] = []
        self.vars: List[Tensor] = []
        self.factors: List[float] = []

    def add(self, bn: nn.BatchNorm2d, mean: Tensor, var: Tensor, n: int):
        if bn.track_running_stats and bn.running_mean is not None:
            self.mods.append(bn)
            self.means.append(mean)
            self.vars.append(var)
            self.factors.append(n / max(1, n - 1))

    @torch.no_grad()
    def flush(self):
        """ONE C-ABI call (mcl_bn_running_update, csrc/step_misc.hip: 64 layers per launch, the pointer table by value in
        the kernel arguments) instead of seven torch._foreach launches."""
        if not self.mods:
            return
        n = len(self.mods)
        import ctypes as C
        vp = C.c_void_p * n
        rm = vp(*[bn.running_mean.data_ptr() for bn in self.mods])
        rv = vp(*[bn.running_var.data_ptr() for bn in self.mods])
        mean = vp(*[t.data_ptr() for t in self.means])
        var = vp(*[t.data_ptr() for t in self.vars])
        nbt = vp(*[(bn.num_batches_tracked.data_ptr() if bn.num_batches_tracked is not None else None) for bn in self.mods])
        cs = (C.c_int32 * n)(*[bn.running_mean.numel() for bn in self.mods])
        fac = (C.c_float * n)(*self.factors)
        # nn.BatchNorm2d(momentum=None) means a cumulative average: 1 / num_batches_tracked -- not used by DenseNet; keep
        # torch's default 0.1 semantics explicit
        mom = (C.c_float * n)(*[(bn.momentum if bn.momentum is not None else 0.1) for bn in self.mods])
        for bn, t in zip(self.mods, self.means):
            if (bn.running_mean.dtype != torch.float32 or not bn.running_mean.is_contiguous()
                    or not bn.running_var.is_contiguous() or not t.is_contiguous() or bn.running_mean.device != t.device):
                raise RuntimeError("densenet_fused: BatchNorm running statistics must be contiguous fp32 on the GPU")
        check(_lib.lib().mcl_bn_running_update(n, rm, rv, mean, var, nbt, cs, fac, mom, _stream()), "mcl_bn_running_update")
        self.mods, self.means, self.vars, self.factors = [], [], [], []


class _BlockStats:
    """Per-channel batch statistics of a dense block's concat buffer."""

    def __init__(self, c_total: int, device):
        self.mean = torch.empty(c_total, device=device, dtype=torch.float32)
        self.var = torch.empty(c_total, device=device, dtype=torch.float32)
        self.rstd = torch.empty(c_total, device=device, dtype=torch.float32)
        # the block's concat buffer when the producer of its input (TransitionFn) already wrote x0 into [:, :C0] of it
        self.buf: Optional[Tensor] = None


def _as2d(t: Tensor) -> Tensor:
    B, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, C)


def _conv1x1_fwd(a: Tensor, w: Tensor) -> Tensor:
    """1x1 convolution outside the fused kernel's shapes (fp32 activations): the activation IS the GEMM operand."""
    from . import conv_generic as cg
    return cg.conv_fwd(a, w, 1, 0)


# norm1 + relu1 + conv1 + norm2-statistics of a dense layer as ONE kernel (csrc/dense_conv.hip): the normalised
# input `a` is never materialised (the weight-gradient kernel recomputes it from the concat buffer).


def dense_conv1x1_fwd(x: Tensor, g1: Tensor, b1: Tensor, mean: Tensor, rstd: Tensor, w16: Tensor, eps2: float,
                      zmean: Optional[Tensor], zvar: Optional[Tensor], zrstd: Optional[Tensor]) -> Tensor:
    """z = conv1x1(relu(bn(x)), w16) and z's batch statistics.  x: channel slice of the concat buffer (bf16
    channels-last), w16: (128, C_in, 1, 1) bf16 whose storage is (128, C_in) row-major."""
    px, S, K, ldx = _rows(x)
    B, _, H, W = x.shape
    z = torch.empty((B, 128, H, W), device=x.device, dtype=torch.bfloat16, memory_format=CL)
    L = _lib.lib()
    ws = _ws(L.mcl_dense_conv1x1_workspace_floats(S), x.device)
    nz = (lambda t: None if t is None else t.data_ptr())
    check(L.mcl_dense_conv1x1_fwd(px, ldx, S, K, g1.data_ptr(), b1.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                  w16.data_ptr(), z.data_ptr(), 128, ws.data_ptr(), eps2, nz(zmean),
                                  nz(zvar), nz(zrstd), _stream()), "mcl_dense_conv1x1_fwd")
    return z


# conv1 backward-data + norm1/relu1 backward as two GEMM-recomputing launches (csrc/dense_bwd.hip)


def dense_bn1_bwd(dz: Tensor, w16: Tensor, x: Tensor, g1: Tensor, b1: Tensor, mean: Tensor, rstd: Tensor, gbuf: Tensor,
                  into_param_grads: bool) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """gbuf (the block's gradient buffer slice) += d loss / d x through conv1 <- relu1 <- norm1, and the norm1
    parameter gradients (added straight into g1.grad / b1.grad when ``into_param_grads``)."""
    px, S, C, ldx = _rows(x)
    pg, S2, C2, ldg = _rows(gbuf)
    assert (S2, C2) == (S, C) and dz.is_contiguous(memory_format=CL) and dz.shape[1] == 128
    L = _lib.lib()
    ws = _ws(L.mcl_dense_bn1_bwd_workspace_floats(S, C), x.device)
    if into_param_grads:
        dg, db = g1.grad, b1.grad
    else:
        dg = torch.empty(C, device=x.device, dtype=torch.float32)
        db = torch.empty(C, device=x.device, dtype=torch.float32)
    check(L.mcl_dense_bn1_bwd(dz.data_ptr(), w16.data_ptr(), C, px, ldx, S, g1.data_ptr(), b1.data_ptr(),
                              mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), dg.data_ptr(), db.data_ptr(),
                              int(into_param_grads), pg, ldg, _stream()), "mcl_dense_bn1_bwd")
    return (None, None) if into_param_grads else (dg, db)


def dense_bn1_dx_sums(dz: Tensor, w16: Tensor, x: Tensor, g1: Tensor, b1: Tensor, mean: Tensor, rstd: Tensor, gbuf: Tensor,
                      kprev: Tensor, have_prev: bool, into_param_grads: bool) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """Single-pass form of ``dense_bn1_bwd`` (csrc/dense_bwd.hip, mcl_dense_bn1_dx_sums): gbuf += gamma*rstd*g, minus the
    previous pass's mean terms (``kprev`` (C_total, 2) fp32, when ``have_prev``), and the norm1 parameter gradients in ONE pass
    over (dz, x); the finalize then overwrites kprev with this layer's mean terms."""
    px, S, C, ldx = _rows(x)
    pg, S2, C2, ldg = _rows(gbuf)
    assert (S2, C2) == (S, C) and dz.is_contiguous(memory_format=CL) and dz.shape[1] == 128 and kprev.numel() >= 2 * C
    L = _lib.lib()
    ws = _ws(L.mcl_dense_bn1_bwd_workspace_floats(S, C), x.device)
    if into_param_grads:
        dg, db = g1.grad, b1.grad
    else:
        dg = torch.empty(C, device=x.device, dtype=torch.float32)
        db = torch.empty(C, device=x.device, dtype=torch.float32)
    check(L.mcl_dense_bn1_dx_sums(dz.data_ptr(), w16.data_ptr(), C, px, ldx, S, g1.data_ptr(), b1.data_ptr(),
                                  mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                  int(into_param_grads), kprev.data_ptr(), int(have_prev), pg, ldg, _stream()),
          "mcl_dense_bn1_dx_sums")
    return (None, None) if into_param_grads else (dg, db)


def dense_bn1_fix(buf: Tensor, gbuf: Tensor, c0: int, nc: int, mean: Tensor, rstd: Tensor, kacc: Tensor) -> None:
    """gbuf[:, c0:c0+nc] -= K1 + K2*xhat: the mean terms of the last single-pass layer, for the channels of the concat buffer
    ``buf`` that no later pass covers (mcl_dense_bn1_fix).  mean / rstd / kacc are the block's full arrays."""
    px, S, _, ldx = _rows(buf)
    pg, S2, _, ldg = _rows(gbuf)
    assert S == S2
    check(_lib.lib().mcl_dense_bn1_fix(px, ldx, pg, ldg, S, c0, nc, mean.data_ptr(), rstd.data_ptr(), kacc.data_ptr(),
                                       _stream()), "mcl_dense_bn1_fix")


# Single-pass BatchNorm-1 backward (each layer's mean terms applied one pass late, one pass over (dz, x) less per layer) on
# the maps where the bottleneck weight gradient does not ride on the reduction anyway (below FUSED_BN1_WRW_MIN_PIXELS).  A first
# version deferred ALL layers' mean terms to one correction per channel range: the bf16 buffer then carried up to 24 layers'
# un-subtracted mean components and the final subtraction cancelled them in bf16 (worst parameter-gradient deviation of
# test_cfg4_backbone_256px_accuracy_vs_fp64: 2.1 -> 12.3-13.2; stock bf16 ops 2.6-5.5); an fp32 side accumulator fixed that
# but doubled the read-modify-write bytes (the gain fell from 0.31 to 0.05 ms/step).
# Only on maps of at most 16 x 16 pixels (the 14 x 14 / 7 x 7 blocks; 16 x 16 / 8 x 8 at 256-pixel patches), whatever the
# batch: the one-pass-late subtraction leaves a slightly larger rounding residue ALONG the directions (1, xhat) that every
# later BatchNorm backward annihilates again -- harmless inside the network, but the first block's input gradient feeds norm0
# directly, whose weight has an exactly zero true gradient (the following BatchNorm layers make the loss invariant to its
# scale): at B = 4, where all four blocks would otherwise qualify, the noise on that parameter grew 6x
# (tools/diag_accuracy_batch.py; test_cfg4_backbone_256px_accuracy_vs_fp64's maximum 2.1 -> 12).
USE_BN1_SINGLE_PASS = True
BN1_SINGLE_PASS_MAX_MAP = 256

# Deterministic fusion of the bottleneck weight gradient with the BatchNorm-backward reduction (csrc/wrw_fused.hip): one
# pass over (dz, x) replaces conv1x1_wrw + the reduce launch + its finalize; then the dx pass alone.  It does the least
# total work but puts the weight gradient ON the critical chain of the backward.  Where the side stream hides the weight
# gradients (14 x 14 and 7 x 7 maps, mostly 28 x 28) the shorter chain wins; on the 56 x 56 maps both lanes are
# throughput-bound, the side work costs its full duration anyway, and saving one pass over (dz, x) per layer wins:
# fused from 200 000 pixels up 14.08 / 14.12 ms/step, from 50 000 up 14.15, never 14.28 / 14.36 (interleaved A/B); after the
# side-lane grids were shrunk (csrc/wrw_fused.hip plan()): from 50 000 up 13.59 / 13.60, from 200 000 up 13.70 / 13.76.
USE_FUSED_BN1_WRW = True
FUSED_BN1_WRW_MIN_PIXELS = 50000


def _bn1_wrw_ok(w_param: Tensor) -> bool:
    return (USE_FUSED_BN1_WRW and DIRECT_PARAM_GRADS and _direct_grad_ok(w_param) and w_param.grad.is_contiguous()
            and w_param.shape[0] == 128 and w_param.shape[2:] == (1, 1))


def dense_bn1_wrw(dz: Tensor, w16: Tensor, x: Tensor, g1: Tensor, b1: Tensor, mean: Tensor, rstd: Tensor, w_param: Tensor,
                  into_param_grads: bool):
    """w_param.grad += dz^T relu(bn1(x)); norm1 parameter gradients; returns (dgamma, dbeta, coef) -- coef = the layer's two
    BatchNorm-backward means per channel, what the dx pass(es) apply (mcl_dense_bn1_wrw: Gram partials + fixed-order merge)."""
    px, S, C, ldx = _rows(x)
    assert dz.is_contiguous(memory_format=CL) and dz.shape[1] == 128
    L = _lib.lib()
    ws = _ws(L.mcl_wrw_workspace_floats(S, 128, C), x.device)
    if into_param_grads:
        dg, db = g1.grad, b1.grad
    else:
        dg = torch.empty(C, device=x.device, dtype=torch.float32)
        db = torch.empty(C, device=x.device, dtype=torch.float32)
    coef = torch.empty(2 * C, device=x.device, dtype=torch.float32)
    check(L.mcl_dense_bn1_wrw(dz.data_ptr(), w16.data_ptr(), C, px, ldx, S, g1.data_ptr(), b1.data_ptr(),
                              mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), w_param.grad.data_ptr(), 1,
                              dg.data_ptr(), db.data_ptr(), int(into_param_grads), coef.data_ptr(), _stream()),
          "mcl_dense_bn1_wrw")
    return (None, None, coef) if into_param_grads else (dg, db, coef)


def dense_bn1_dx(dz: Tensor, w16: Tensor, x: Tensor, g1: Tensor, b1: Tensor, mean: Tensor, rstd: Tensor, coef: Tensor,
                 gbuf: Tensor, window=None) -> None:
    """gbuf += d loss / d x of the layer head (mcl_dense_bn1_dx); ``window`` = (c0, nc): only those input channels."""
    px, S, C, ldx = _rows(x)
    pg, S2, C2, ldg = _rows(gbuf)
    assert (S2, C2) == (S, C)
    L = _lib.lib()
    if window is None:
        check(L.mcl_dense_bn1_dx(dz.data_ptr(), w16.data_ptr(), C, px, ldx, S, g1.data_ptr(), b1.data_ptr(),
                                 mean.data_ptr(), rstd.data_ptr(), coef.data_ptr(), pg, ldg, _stream()), "mcl_dense_bn1_dx")
    else:
        c0, nc = window
        check(L.mcl_dense_bn1_dx_window(dz.data_ptr(), w16.data_ptr(), C, c0, nc, px, ldx, S, g1.data_ptr(), b1.data_ptr(),
                                        mean.data_ptr(), rstd.data_ptr(), coef.data_ptr(), pg, ldg, _stream()),
              "mcl_dense_bn1_dx_window")


def dense_bn1_dx_pair(A, B, x: Tensor, mean: Tensor, rstd: Tensor, gbuf: Tensor) -> None:
    """Layers A = l and B = l - 1 (each a tuple (dz, w16, g1, b1, coef)): gbuf[:, :C] += both layers' terms in ONE pass over
    the C channels layer B reads (mcl_dense_bn1_dx_pair); x / gbuf are the [:C] slices."""
    dzA, wA, gA, bA, cA = A
    dzB, wB, gB, bB, cB = B
    px, S, C, ldx = _rows(x)
    pg, S2, C2, ldg = _rows(gbuf)
    assert (S2, C2) == (S, C) and wA.shape[1] >= C and wB.shape[1] == C
    check(_lib.lib().mcl_dense_bn1_dx_pair(dzA.data_ptr(), wA.data_ptr(), wA.shape[1], gA.data_ptr(), bA.data_ptr(), cA.data_ptr(),
                                           dzB.data_ptr(), wB.data_ptr(), gB.data_ptr(), bB.data_ptr(), cB.data_ptr(), C, px, ldx, S,
                                           mean.data_ptr(), rstd.data_ptr(), pg, ldg, _stream()), "mcl_dense_bn1_dx_pair")


def dense_bn1_wrw_dx(dz: Tensor, w16: Tensor, x: Tensor, g1: Tensor, b1: Tensor, mean: Tensor, rstd: Tensor, gbuf: Tensor,
                     w_param: Tensor, into_param_grads: bool) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """w_param.grad += dz^T relu(bn1(x)); norm1 parameter gradients; gbuf += d loss / d x -- two C-ABI calls
    (mcl_dense_bn1_wrw: Gram partials + fixed-order merge; mcl_dense_bn1_dx)."""
    dg, db, coef = dense_bn1_wrw(dz, w16, x, g1, b1, mean, rstd, w_param, into_param_grads)
    dense_bn1_dx(dz, w16, x, g1, b1, mean, rstd, coef, gbuf)
    return dg, db


# Round 6: the dx passes of two consecutive layers of a 56 x 56 / 28 x 28 block as ONE pass over the channels both read (x and
# the gradient buffer read once, the buffer written once; layer l's term on the 32 channels layer l - 1 produced goes first, in a
# windowed launch).  MCL_BN1_PAIR=0: one dx pass per layer (A/B).
USE_BN1_PAIR = os.environ.get("MCL_BN1_PAIR", "1") != "0"


# conv2 (3x3) backward-data + norm2/relu2 backward (csrc/dense_bwd.hip): dy is read in place from the gradient buffer


# The 32-channel mean-term correction of the single-pass BatchNorm-1 backward (mcl_dense_bn1_fix: a 5 us launch in front of
# every 3x3 backward-data kernel of the 14 x 14 / 7 x 7 blocks, 38 per step on the critical chain) folded into that kernel's
# dy staging.  MCL_FOLD_BN1_FIX=0: the separate launch (A/B; bit-identical results).
FOLD_BN1_FIX = True


def _c3_flat_kernel(W: int) -> bool:
    """True when the 3x3 backward-data of a W-wide map runs the flat-tile kernel (csrc/dense_bwd.hip bwd_rows_applicable)."""
    if False:
        return True
    return W < 17 or W > 150


def dense_conv3x3_bwd(dy: Tensor, w16: Tensor, z: Tensor, g2: Tensor, b2: Tensor, m2: Tensor, r2: Tensor,
                      into_param_grads: bool, fix=None) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor], Optional[Tensor]]:
    """(dz, dgamma2, dbeta2, dyc): gradient of the loss w.r.t. the bottleneck output z through conv2 <- relu2 <- norm2.
    ``fix`` = (x, mean, rstd, k): the layer's 32 output channels of the concat buffer, their statistics and the previous
    single-pass layer's mean terms -- dy is corrected while it is staged (mcl_dense_conv3x3_bwd_fix) and the corrected copy
    comes back as ``dyc`` (B, 32, H, W) for the weight-gradient kernel; else dyc is None."""
    B, C, H, W = z.shape
    pd, S, Co, lddy = _rows(dy)
    assert C == 128 and Co == 32 and S == B * H * W and z.is_contiguous(memory_format=CL)
    L = _lib.lib()
    ws = _ws(L.mcl_dense_conv3x3_bwd_workspace_floats(S), z.device)
    scratch = torch.empty_like(z, memory_format=CL)
    dz = torch.empty_like(z, memory_format=CL)
    if into_param_grads:
        dg, db = g2.grad, b2.grad
    else:
        dg = torch.empty(C, device=z.device, dtype=torch.float32)
        db = torch.empty(C, device=z.device, dtype=torch.float32)
    dyc = None
    if fix is not None:
        xf, fm, fr, fk = fix
        pxf, S2, C2, ldxf = _rows(xf)
        assert (S2, C2) == (S, 32) and fk.is_contiguous() and fk.numel() == 64
        dyc = torch.empty((B, 32, H, W), device=z.device, dtype=z.dtype, memory_format=CL)
        check(L.mcl_dense_conv3x3_bwd_fix(pd, lddy, S, H, W, w16.data_ptr(), z.data_ptr(), g2.data_ptr(), b2.data_ptr(),
                                          m2.data_ptr(), r2.data_ptr(), ws.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                          int(into_param_grads), scratch.data_ptr(), dz.data_ptr(), pxf, ldxf,
                                          fm.data_ptr(), fr.data_ptr(), fk.data_ptr(), dyc.data_ptr(), _stream()),
              "mcl_dense_conv3x3_bwd_fix")
    else:
        check(L.mcl_dense_conv3x3_bwd(pd, lddy, S, H, W, w16.data_ptr(), z.data_ptr(), g2.data_ptr(), b2.data_ptr(),
                                      m2.data_ptr(), r2.data_ptr(), ws.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                      int(into_param_grads), scratch.data_ptr(), dz.data_ptr(), _stream()),
              "mcl_dense_conv3x3_bwd")
    return (dz, None, None, dyc) if into_param_grads else (dz, dg, db, dyc)


# norm2 + relu2 + conv2 (3x3) + the new feature map's statistics as ONE kernel writing into the concat buffer


def dense_conv3x3_fwd(z: Tensor, g2: Tensor, b2: Tensor, m2: Tensor, r2: Tensor, w16: Tensor, out: Tensor, eps: float,
                      ymean: Optional[Tensor], yvar: Optional[Tensor], yrstd: Optional[Tensor]) -> None:
    """out (a 32-channel slice of the concat buffer) = conv3x3(relu(bn2(z)), w16), plus its batch statistics."""
    B, C, H, W = z.shape
    po, S, Co, ldo = _rows(out)
    assert C == 128 and Co == 32 and z.is_contiguous(memory_format=CL) and S == B * H * W
    L = _lib.lib()
    ws = _ws(L.mcl_dense_conv3x3_workspace_floats(S), z.device)
    nz = (lambda t: None if t is None else t.data_ptr())
    check(L.mcl_dense_conv3x3_fwd(z.data_ptr(), S, H, W, g2.data_ptr(), b2.data_ptr(), m2.data_ptr(), r2.data_ptr(),
                                  w16.data_ptr(), po, ldo, ws.data_ptr(), eps, nz(ymean), nz(yvar),
                                  nz(yrstd), _stream()), "mcl_dense_conv3x3_fwd")


def _grad_target_khwc(w_param: Tensor) -> Tuple[Tensor, bool]:
    """The weight-gradient kernels write (C_out, kh, kw, C_in)-contiguous fp32.  A channels-last parameter's .grad (what
    train.py / bench.py hold: ``model.to(memory_format=channels_last)``) IS that layout and is accumulated in place; for a
    default-contiguous (NCHW) parameter the kernel fills a temporary that ``_grad_finish_khwc`` adds into .grad."""
    g = w_param.grad
    if g.permute(0, 2, 3, 1).is_contiguous():
        return g, True
    co, ci, kh, kw = w_param.shape
    return torch.empty((co, kh, kw, ci), device=g.device, dtype=torch.float32), False


def _grad_finish_khwc(w_param: Tensor, tgt: Tensor, in_place: bool) -> None:
    if not in_place:
        w_param.grad.add_(tgt.permute(0, 3, 1, 2))


def dense_conv3x3_wrw(dy: Tensor, z: Tensor, g2: Tensor, b2: Tensor, m2: Tensor, r2: Tensor, w_param: Tensor) -> bool:
    """Adds the 3x3 weight gradient (a2 = relu(bn2(z)) recomputed in-kernel) straight into ``w_param.grad``.
    Returns False (nothing done) when the parameter cannot take a dense fp32 .grad or the operands are not bf16."""
    if not _wrw3_direct_ok(w_param, z, dy):
        return False
    B, C, H, W = z.shape
    pd, S, Co, lddy = _rows(dy)
    assert Co == 32 and S == B * H * W
    L = _lib.lib()
    tgt, in_place = _grad_target_khwc(w_param)
    # the side stream has its own workspace (keyed by stream in _ws): no aliasing with the main chain's
    ws = _ws(L.mcl_dense_conv3x3_wrw_workspace_floats(S), z.device)
    check(L.mcl_dense_conv3x3_wrw_det(pd, lddy, z.data_ptr(), S, H, W, g2.data_ptr(), b2.data_ptr(), m2.data_ptr(),
                                      r2.data_ptr(), ws.data_ptr(), tgt.data_ptr(), int(in_place), _stream()),
          "mcl_dense_conv3x3_wrw_det")
    _grad_finish_khwc(w_param, tgt, in_place)
    return True


def _wrw3_direct_ok(w_param: Tensor, z: Tensor, dy: Tensor) -> bool:
    """dense_conv3x3_wrw's precondition (it accumulates into w_param.grad)."""
    return (DIRECT_PARAM_GRADS and _direct_grad_ok(w_param) and tuple(w_param.shape) == (32, 128, 3, 3)
            and dy.dtype == torch.bfloat16 and z.dtype == torch.bfloat16 and z.is_contiguous(memory_format=CL))


def _fused_3x3_ok(z: Tensor, w16: Tensor) -> bool:
    return (z.dtype == torch.bfloat16 and w16.dtype == torch.bfloat16
            and tuple(w16.shape) == (32, 128, 3, 3) and w16.permute(0, 2, 3, 1).is_contiguous()
            and z.shape[1] == 128 and z.shape[3] <= 150 and z.is_contiguous(memory_format=CL))


def _fused_1x1_ok(x: Tensor, w16: Tensor) -> bool:
    return (x.dtype == torch.bfloat16 and w16.dtype == torch.bfloat16 and w16.shape[0] == 128
            and w16.shape[1] % 8 == 0 and w16.shape[1] <= 1024 and w16.shape[2:] == (1, 1)
            and w16.permute(0, 2, 3, 1).is_contiguous())


def conv1x1_wrw(dz: Tensor, a: Tensor, w_param: Tensor, bn=None) -> Optional[Tensor]:
    """Weight gradient of a 1x1 convolution (csrc/wrw_fused.hip, atomics-free).  Adds straight into ``w_param.grad``
    when it is a dense fp32 tensor (returns None), else returns a fresh fp32 gradient.  ``bn`` = (gamma, beta,
    mean, rstd): ``a`` is then the layer INPUT (concat-buffer slice) and relu(bn(a)) is recomputed in-kernel."""
    pz, S, M, ldz = _rows(dz)
    pa, S2, N, lda = _rows(a)
    assert S == S2 and dz.dtype == a.dtype == torch.bfloat16
    if DIRECT_PARAM_GRADS and _direct_grad_ok(w_param) and w_param.grad.is_contiguous():
        tgt, ret = w_param.grad, None
    else:
        tgt = torch.zeros((M, N, 1, 1), device=dz.device, dtype=torch.float32)
        ret = tgt
    g_, b_, m_, r_ = (t.data_ptr() for t in bn) if bn is not None else (None, None, None, None)
    L = _lib.lib()
    ws = _ws(L.mcl_wrw_workspace_floats(S, min(M, 128), N), dz.device)
    check(L.mcl_conv1x1_wrw_det(pz, ldz, pa, lda, g_, b_, m_, r_, ws.data_ptr(), tgt.data_ptr(), 1, S, M, N,
                                _stream()), "mcl_conv1x1_wrw_det")
    return ret


def _conv1x1_bwd(dz: Tensor, a: Tensor, w: Tensor, w_param: Tensor):
    """(da, dw) of a 1x1 convolution on the generic path; dw is None when it was added straight into ``w_param.grad``."""
    return _conv_bwd(dz, a, w, w_param, 0)


def _same_order(a: Tensor, b: Tensor) -> bool:
    """Same shape and same storage order (strides compared on dims of extent > 1 only), both dense."""
    if a.shape != b.shape:
        return False
    sa = [s for s, n in zip(a.stride(), a.shape) if n > 1]
    sb = [s for s, n in zip(b.stride(), b.shape) if n > 1]
    dense = sorted(sa, reverse=True) and True
    return sa == sb and dense


def _wgrad(w: Tensor, dw: Tensor) -> Optional[Tensor]:
    """Weight gradient hand-over: add into an existing dense fp32 .grad (one mixed-dtype add kernel) or
    return it to autograd in the parameter's dtype."""
    if DIRECT_PARAM_GRADS and _direct_grad_ok(w) and _same_order(dw, w.grad) and dw.dtype in (
            torch.bfloat16, torch.float32):
        # (a mixed-dtype torch add_ on the channels-last strided view costs 45 us per weight)
        check(_lib.lib().mcl_accum_into_f32(w.grad.data_ptr(), dw.data_ptr(), dw.numel(), _dt(dw), _stream()),
              "mcl_accum_into_f32")
        return None
    return dw.to(w.dtype)


# Optional provider of low-precision weight copies: FusedAdam keeps every parameter in one flat fp32 buffer, so
# ONE cast kernel per step yields a flat bf16 shadow whose views replace ~120 per-weight cast kernels.
_weight_provider = None


def set_weight_provider(fn) -> None:
    global _weight_provider
    _weight_provider = fn


def cast_dense_bf16(w: Tensor) -> Optional[Tensor]:
    """bf16 copy of a dense fp32 GPU tensor with the SAME strides, on this library's cast kernel (inference without an attached
    FusedAdam has no flat shadow: ~120 per-weight ATen casts per forward otherwise).  None when the tensor is not eligible."""
    w = w.detach()
    if not (w.is_cuda and w.dtype == torch.float32 and w.numel() % 8 == 0 and w.numel() > 0 and w.data_ptr() % 16 == 0):
        return None
    dense = w.is_contiguous() or (w.dim() == 4 and w.is_contiguous(memory_format=CL))
    if not dense:
        return None
    out = torch.empty_strided(w.shape, w.stride(), device=w.device, dtype=torch.bfloat16)
    check(_lib.lib().mcl_cast_f32_to_bf16(w.data_ptr(), w.numel(), out.data_ptr(), w.numel(), 1, w.numel(), _stream()),
          "mcl_cast_f32_to_bf16")
    return out


def _weight(w: Tensor, dt: torch.dtype) -> Tensor:
    if w.dtype == dt:
        return w if w.is_contiguous(memory_format=CL) else w.contiguous(memory_format=CL)
    if _weight_provider is not None:
        v = _weight_provider(w, dt)
        if v is not None:
            return v if v.is_contiguous(memory_format=CL) else v.contiguous(memory_format=CL)
    if dt == torch.bfloat16 and w.dim() == 4 and w.is_contiguous(memory_format=CL):
        v = cast_dense_bf16(w)
        if v is not None:
            return v
    return w.to(dtype=dt, memory_format=CL)


def _conv_bwd(dy: Tensor, x: Tensor, w: Tensor, w_param: Tensor, padding: int, cols: Optional[Tensor] = None):
    """(dx, dw) of a stride-1 convolution on the generic path (conv_generic.py: GEMM + col2im, split-K weight gradient);
    dw is None when it was accumulated straight into ``w_param.grad``, else an fp32 tensor shaped like the parameter.
    ``cols``: the forward's unfolded patches when it kept them."""
    from . import conv_generic as cg
    # (a channel slice of the gradient buffer is read in place: the GEMMs / im2col take a row stride)
    try:
        _rows(dy)
    except RuntimeError:
        dy = dense_cl(dy)
    dw = cg.conv_bwd_weight(dy, x, w_param, 1, padding, cols)
    dx = cg.conv_bwd_data(dy, w, x.shape, 1, padding)
    return dx, dw


# The two weight-gradient kernels of a layer are independent of its data-gradient chain (they only add into .grad):
# issue them on a side stream so they overlap the latency-bound backward-data / BatchNorm-backward launches (under
# HIP-graph capture this becomes a parallel branch of the graph).  Joined at the end of every layer.
USE_SIDE_STREAM = True             # (bench.py switches it off for its per-kernel timing pass)
_side_streams = {}


def _side_stream(device) -> torch.cuda.Stream:
    s = _side_streams.get(device.index)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _side_streams[device.index] = s
    return s


# Deferred joins of the side stream.  Every cross-stream edge of the captured graph costs the WAITING stream ~16 us
# (tools/trace_gaps.py), so the main chain never waits per kernel: side work is forked off with an event, the tensors
# it reads are parked here (so that the allocator cannot hand their memory out again), and the main stream joins
# once per dense block / at the stem.
_side_parked: dict = {}
JOIN_MIN_PIXELS = 300000


def _side_park(device, *tensors) -> None:
    _side_parked.setdefault(device.index, []).extend(tensors)


def _side_join(device) -> None:
    if _side_pending.get(device.index):
        # deferred side work that no later fork picked up (the block was the last one of this backward): issue it now
        main = torch.cuda.current_stream(device)
        side = _side_stream(device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            _run_side_pending(device)
    parked = _side_parked.get(device.index)
    if parked:
        torch.cuda.current_stream(device).wait_stream(_side_stream(device))
        parked.clear()


# Side work whose only dependency is "everything the main stream has issued so far", deferred to the NEXT fork the backward
# makes anyway.  The persistent dense-block backward is ONE kernel followed by 32 weight-gradient launches: forking the side
# stream right behind that kernel (one cross-stream edge out of a 0.9 ms node) made the replayed step graph lose its two-lane
# execution -- the spot branch's backward, which precedes those launches on the side stream, then ran 4.5 ms late
# (profiles/r05_persistent_bwd_lanes.txt).  Riding on the next block's first per-layer fork keeps the graph's edge structure
# what it was.
_side_pending: dict = {}


def _defer_to_side(device, fn) -> None:
    _side_pending.setdefault(device.index, []).append(fn)


def _run_side_pending(device) -> None:
    """Called with the side stream current, after it has waited for an event the main stream recorded later than every
    deferred job's inputs."""
    jobs = _side_pending.get(device.index)
    if jobs:
        for fn in jobs:
            fn()
        jobs.clear()


# Test instrumentation (tests/test_layerwise_gpu.py): a list that receives, per dense block, the tensors every fused layer
# kernel of the block consumed and produced (forward: concat buffer, statistics, z; backward: incoming gradient buffer, dz
# per layer, final gradient buffer), so that each kernel can be checked against an fp64 evaluation of exactly its inputs
# at the benched shapes.  None (the default) = nothing is recorded.
CAPTURE_BLOCKS: Optional[list] = None
# The same for everything around the dense blocks (stem convolution, norm0 + pool0, the transitions, norm5 + global pool):
# a list of dicts {"kind": ..., tensors consumed / produced forward and backward}.
CAPTURE_MISC: Optional[list] = None


USE_BLOCK_PERSISTENT = os.environ.get("MCL_BLOCK_PERSIST", "1") != "0"
# test hook (tests/test_dense_block_gpu.py): bound of the seam polls of the persistent kernels; 0 = the library's default (2^19)
SEAM_MAX_SPINS = 0
# profiling hook (tools/bench_dense_block.py): a uint64 device tensor of B*L*8 words the persistent launches fill with in-kernel
# phase stamps; None (the default) = off.  Passed per call: the library keeps no state.
BLOCK_STAMPS: Optional[Tensor] = None


def _stamps_ptr():
    return BLOCK_STAMPS.data_ptr() if BLOCK_STAMPS is not None else None
_cu_count: dict = {}


def _cus(device) -> int:
    n = _cu_count.get(device.index)
    if n is None:
        n = _cu_count[device.index] = int(torch.cuda.get_device_properties(device).multi_processor_count)
    return n


def _block_persistent_ok(buf: Tensor, params, growth: int, L: int, C0: int, dt: torch.dtype) -> bool:
    B, Ct, H, W = buf.shape
    if not (USE_BLOCK_PERSISTENT and H == 7 and W == 7 and growth == 32 and dt == torch.bfloat16 and L <= 24 and Ct <= 1024
            and C0 % 32 == 0 and buf.is_cuda and B <= _cus(buf.device)):
        return False
    for l in range(L):
        w1, w2 = params[6 * l + 2], params[6 * l + 5]
        if tuple(w1.shape) != (128, C0 + l * growth, 1, 1) or tuple(w2.shape) != (32, 128, 3, 3):
            return False
    return True


def block_persistent_error(device) -> bool:
    """True if a persistent dense-block launch on ``device`` gave up waiting at a seam since the last check (host sync).  The
    production path raises instead: ``ops.check_device_errors`` (train.train, TrainStep.check_errors / its polling)."""
    from . import ops as _ops
    t = _ops.block_seam_error_flag(device)
    bad = int(t.item()) != 0
    if bad:
        t.zero_()
    return bad


USE_BLOCK_PERSISTENT_BWD = os.environ.get("MCL_BLOCK_PERSIST_BWD", "1") != "0"


def _block_bwd_persistent_ok(buf: Tensor, gbuf: Tensor, params, L: int) -> bool:
    """The persistent backward adds every BatchNorm gradient straight into ``.grad`` and hands dz / dy' to the direct weight
    gradient kernels: all parameters of the block must take the direct path."""
    if not (USE_BLOCK_PERSISTENT_BWD and DIRECT_PARAM_GRADS and gbuf.dtype == torch.bfloat16
            and gbuf.is_contiguous(memory_format=CL) and tuple(gbuf.shape) == tuple(buf.shape)):
        return False
    for l in range(L):
        g1, b1, w1, g2, b2, w2 = params[6 * l: 6 * l + 6]
        if not all(_direct_grad_ok(p) for p in (g1, b1, g2, b2, w2)):
            return False
    return True


def dense_block_bwd_persistent(buf: Tensor, gbuf: Tensor, params, wcast, zs, stats: "_BlockStats", bn2_stats, C0: int, L: int):
    """csrc/dense_block.hip dense_block_bwd_kernel: the data-gradient chain of all L layers in one launch.  gbuf[:, :C0]
    receives the block-input gradient; norm1 / norm2 gradients are accumulated into ``.grad``; returns (dz list, dy' list)."""
    import ctypes as C
    B, Ct, H, W = buf.shape
    dev = buf.device
    Lb = _lib.lib()
    dzs = [torch.empty((B, 128, H, W), device=dev, dtype=torch.bfloat16, memory_format=CL) for _ in range(L)]
    dycs = [torch.empty((B, 32, H, W), device=dev, dtype=torch.bfloat16, memory_format=CL) for _ in range(L)]
    n1 = [128 * (C0 + 32 * l) for l in range(L)]
    n2 = 4 * 18 * 64 * 8
    packed = torch.empty(sum(n1) + L * n2, device=dev, dtype=torch.bfloat16)
    o1, o = [], 0
    for n in n1:
        o1.append(o)
        o += n
    o2 = [o + l * n2 for l in range(L)]
    vp = C.c_void_p * L
    check(Lb.mcl_dense_block_pack_bwd(vp(*[wcast[2 * l].data_ptr() for l in range(L)]),
                                      vp(*[wcast[2 * l + 1].data_ptr() for l in range(L)]),
                                      vp(*[packed.data_ptr() + 2 * o1[l] for l in range(L)]),
                                      vp(*[packed.data_ptr() + 2 * o2[l] for l in range(L)]), L, C0, _stream()),
          "mcl_dense_block_pack_bwd")
    ptrs = []
    for l in range(L):
        g1, b1, _, g2, b2, _ = params[6 * l: 6 * l + 6]
        m2, v2, r2 = bn2_stats[l]
        ptrs += [g1.data_ptr(), b1.data_ptr(), packed.data_ptr() + 2 * o1[l], g2.data_ptr(), b2.data_ptr(),
                 packed.data_ptr() + 2 * o2[l], zs[l].data_ptr(), m2.data_ptr(), r2.data_ptr(), dzs[l].data_ptr(),
                 dycs[l].data_ptr(), g1.grad.data_ptr(), b1.grad.data_ptr(), g2.grad.data_ptr(), b2.grad.data_ptr()]
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    nbytes = Lb.mcl_dense_block_bwd_workspace_bytes(B, L)
    ws = _ws((nbytes + 255 + 3) // 4 + 64, dev)
    base = (ws.data_ptr() + 255) & ~255
    from . import ops as _ops
    err = _ops.block_seam_error_flag(dev)
    px, S, C_, ld = _rows(buf)
    pg, S2, C2, ldg = _rows(gbuf)
    if ld != Ct or ldg != Ct:
        raise RuntimeError("dense_block_bwd_persistent: the concat and gradient buffers must be dense channels-last")
    check(Lb.mcl_dense_block_bwd(px, pg, B, H, W, Ct, C0, L, arr, stats.mean.data_ptr(), stats.rstd.data_ptr(), base,
                                 err.data_ptr(), SEAM_MAX_SPINS, _stamps_ptr(), _stream()), "mcl_dense_block_bwd")
    return dzs, dycs


def dense_block_fwd_persistent(buf: Tensor, params, wcast, stats: "_BlockStats", bn2_stats, C0: int, L: int, eps1: float,
                               eps2: float):
    """csrc/dense_block.hip: all L layers of the block in one launch; fills buf[:, C0:], stats[C0:], bn2_stats, returns
    the per-layer z tensors (saved for the backward)."""
    import ctypes as C
    B, Ct, H, W = buf.shape
    dev = buf.device
    Lb = _lib.lib()
    zs = [torch.empty((B, 128, H, W), device=dev, dtype=torch.bfloat16, memory_format=CL) for _ in range(L)]
    # conv1 weights in the kernel's streaming order (one launch per call: the weights change every step)
    sizes = [128 * (C0 + 32 * l) for l in range(L)]
    packed = torch.empty(sum(sizes), device=dev, dtype=torch.bfloat16)
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o += n
    for l in range(L):
        w1c, w2c = wcast[2 * l], wcast[2 * l + 1]
        if not (w1c.permute(0, 2, 3, 1).is_contiguous() and w2c.permute(0, 2, 3, 1).is_contiguous()):
            raise RuntimeError("dense_block_fwd_persistent: weights must be channels-last / (N, K) row-major")
    src = (C.c_void_p * L)(*[wcast[2 * l].data_ptr() for l in range(L)])
    dst = (C.c_void_p * L)(*[packed.data_ptr() + 2 * offs[l] for l in range(L)])
    check(Lb.mcl_dense_block_pack_w1(src, dst, L, C0, _stream()), "mcl_dense_block_pack_w1")
    ptrs = []
    for l in range(L):
        g1, b1, _, g2, b2, _ = params[6 * l: 6 * l + 6]
        w2c = wcast[2 * l + 1]
        m2, v2, r2 = bn2_stats[l]
        ptrs += [g1.data_ptr(), b1.data_ptr(), packed.data_ptr() + 2 * offs[l], g2.data_ptr(), b2.data_ptr(), w2c.data_ptr(),
                 zs[l].data_ptr(), m2.data_ptr(), v2.data_ptr(), r2.data_ptr()]
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    nbytes = Lb.mcl_dense_block_fwd_workspace_bytes(B, L)
    ws = _ws((nbytes + 255 + 3) // 4 + 64, dev)
    base = (ws.data_ptr() + 255) & ~255
    from . import ops as _ops
    err = _ops.block_seam_error_flag(dev)
    px, S, C_, ld = _rows(buf)
    if ld != Ct:
        raise RuntimeError("dense_block_fwd_persistent: the concat buffer must be dense channels-last")
    check(Lb.mcl_dense_block_fwd(px, B, H, W, Ct, C0, L, arr, eps1, eps2, stats.mean.data_ptr(), stats.var.data_ptr(),
                                 stats.rstd.data_ptr(), base, err.data_ptr(), SEAM_MAX_SPINS, _stamps_ptr(), _stream()), "mcl_dense_block_fwd")
    return zs


class DenseBlockFn(torch.autograd.Function):
    """A whole torchvision ``_DenseBlock`` (forward AND hand-scheduled backward).

    inputs: x0 (B, C0, H, W) channels-last; meta = (stats: _BlockStats, eps1 list, eps2 list, growth);
    then per layer: norm1.weight, norm1.bias, conv1.weight, norm2.weight, norm2.bias, conv2.weight.
    output: the concat buffer (B, C0 + L*growth, H, W); ``stats`` is filled as a side effect.
    """

    @staticmethod
    def forward(ctx, x0, meta, *params):
        stats, eps1, eps2, growth, bn2_stats = meta[:5]
        prefilled = len(meta) > 5 and meta[5]       # stats[:C0] already hold x0's statistics (TransitionFn)
        # no transition in front of it (the network's first dense block), or a backward segment ends with this block (its input
        # is cut: the segment's gradient range must be final, and under capture every forked stream joined, when its graph
        # ends): the block joins the weight-gradient side stream at the end of its backward
        ctx.first_block = (not prefilled) or (len(meta) > 6 and bool(meta[6]))
        L = len(params) // 6
        B, C0, H, W = x0.shape
        Ct = C0 + L * growth
        dev, dt = x0.device, x0.dtype
        pre = getattr(stats, "buf", None) if prefilled else None
        if (pre is not None and tuple(pre.shape) == (B, Ct, H, W) and pre.dtype == dt and x0.data_ptr() == pre.data_ptr()
                and x0.stride() == pre[:, :C0].stride()):
            # the transition's convolution wrote x0 straight into the buffer: no copy (a fresh tensor object on the same
            # storage: x0 is a view of `pre` and this function's output must not be that view's base object)
            buf, adopted = pre.detach(), True
        else:
            buf, adopted = torch.empty((B, Ct, H, W), device=dev, dtype=dt, memory_format=CL), False
            x0 = dense_cl(x0)
        if adopted:
            pass
        elif prefilled:
            buf[:, :C0].copy_(x0)
        else:
            bn_stats(x0, stats.mean[:C0], stats.var[:C0], stats.rstd[:C0], eps1[0], copy_out=buf[:, :C0])
        saved = []
        wcast = []
        kept_cols = {}                 # layer -> unfolded 3x3 input of the generic (fp32) path, kept for the weight gradient
        ctx.persistent = False
        if _block_persistent_ok(buf, params, growth, L, C0, dt):
            # 7 x 7 maps: the whole block as ONE persistent launch (csrc/dense_block.hip) -- one workgroup per image, the
            # batch statistics exchanged through two all-to-all seams per layer instead of four dependent launches per layer
            wcast = [_weight(params[6 * l + k], dt) for l in range(L) for k in (2, 5)]
            zs = dense_block_fwd_persistent(buf, params, wcast, stats, bn2_stats, C0, L, eps1[0], eps2[0])
            for l in range(L):
                saved += [buf.new_empty(0), zs[l], buf.new_empty(0)]
            L_done = L
            ctx.persistent = True
        else:
            L_done = 0
        for l in range(L_done, L):
            g1, b1, w1, g2, b2, w2 = params[6 * l: 6 * l + 6]
            cin = C0 + l * growth
            w1c = _weight(w1, dt)
            m2, v2, r2 = bn2_stats[l]
            if _fused_1x1_ok(buf, w1c):
                a = None      # never materialised; the backward recomputes relu(bn1(.)) from buf where needed
                z = dense_conv1x1_fwd(buf[:, :cin], g1, b1, stats.mean[:cin], stats.rstd[:cin], w1c, eps2[l], m2, v2, r2)
            else:
                a = torch.empty((B, cin, H, W), device=dev, dtype=dt, memory_format=CL)
                bn_act_fwd(buf[:, :cin], g1, b1, stats.mean[:cin], stats.rstd[:cin], True, a)
                z = _conv1x1_fwd(a, w1c)
                bn_stats(z, m2, v2, r2, eps2[l])
            w2c = _weight(w2, dt)
            c1 = cin + growth
            # eps of the NEXT consumer's norm1 is the same module default everywhere (1e-5); rstd is
            # stored for eps1[min(l+1, L-1)] -- all equal in torchvision's DenseNet
            if growth == 32 and _fused_3x3_ok(z, w2c):
                a2 = None     # never materialised in the forward; recomputed from z in the backward
                dense_conv3x3_fwd(z, g2, b2, m2, r2, w2c, buf[:, cin:c1], eps1[min(l + 1, L - 1)],
                                  stats.mean[cin:c1], stats.var[cin:c1], stats.rstd[cin:c1])
            else:
                from . import conv_generic as cg
                a2 = torch.empty_like(z, memory_format=CL)
                bn_act_fwd(z, g2, b2, m2, r2, True, a2)
                _, cols2 = cg.conv_fwd(a2, w2c, 1, 1, out=buf[:, cin:c1], want_cols=True)   # written straight into the concat buffer
                kept_cols[l] = cols2
                bn_stats(buf[:, cin:c1], stats.mean[cin:c1], stats.var[cin:c1], stats.rstd[cin:c1], eps1[min(l + 1, L - 1)])
            saved += [a if a is not None else buf.new_empty(0), z, a2 if a2 is not None else buf.new_empty(0)]
            wcast += [w1c, w2c]
        ctx.save_for_backward(buf, *saved, *wcast)
        ctx.kept_cols = kept_cols
        ctx.params = params             # Parameter objects (for direct .grad accumulation)
        ctx.meta = (stats, growth, bn2_stats, L, C0)
        ctx.cap = None
        if CAPTURE_BLOCKS is not None:
            ctx.cap = {"buf": buf, "C0": C0, "growth": growth, "params": params, "wcast": wcast,
                       "mean": stats.mean, "rstd": stats.rstd, "var": stats.var, "z": saved[1::3], "bn2": bn2_stats}
            CAPTURE_BLOCKS.append(ctx.cap)
        return buf

    @staticmethod
    def backward(ctx, gbuf):
        stats, growth, bn2_stats, L, C0 = ctx.meta
        t = ctx.saved_tensors
        buf = t[0]
        params = ctx.params
        saved = t[1: 1 + 3 * L]
        wcast = t[1 + 3 * L:]
        # the incoming gradient is produced by our own BNActFn for the block's single consumer: accumulate
        # in place into it (no clone) when it is already a dense channels-last tensor
        if not gbuf.is_contiguous(memory_format=CL):
            gbuf = dense_cl(gbuf)
        if ctx.cap is not None:
            ctx.cap["gin"] = gbuf.clone(memory_format=CL)
            ctx.cap["dz"] = [None] * L
            ctx.cap["gbuf"] = gbuf
        grads = [None] * (6 * L)
        stamp(f"bwd block {buf.shape[2]}x{buf.shape[3]} start (main)")
        if getattr(ctx, "persistent", False) and _block_bwd_persistent_ok(buf, gbuf, params, L):
            # 7 x 7 maps: the block's whole data-gradient chain as ONE persistent launch (csrc/dense_block.hip); the two weight
            # gradients of every layer follow on the side stream from the dz / dy' tensors it wrote
            zs = [saved[3 * l + 1] for l in range(L)]
            dzs, dycs = dense_block_bwd_persistent(buf, gbuf, params, wcast, zs, stats, bn2_stats, C0, L)
            if ctx.cap is not None:
                ctx.cap["dz"] = list(dzs)
                ctx.cap["dyc"] = list(dycs)
            dev = buf.device
            gw1s = {}

            def _weight_grads():
                for l in range(L - 1, -1, -1):
                    g1, b1, w1, g2, b2, w2 = params[6 * l: 6 * l + 6]
                    cin = C0 + l * growth
                    m2, v2, r2 = bn2_stats[l]
                    ok = dense_conv3x3_wrw(dycs[l], zs[l], g2, b2, m2, r2, w2)
                    assert ok
                    gw1s[l] = conv1x1_wrw(dzs[l], buf[:, :cin], w1, bn=(g1, b1, stats.mean[:cin], stats.rstd[:cin]))
                if STAMPS:
                    stamp(f"bwd block {buf.shape[2]}x{buf.shape[3]} weight gradients done (side)")

            joins = buf.shape[0] * buf.shape[2] * buf.shape[3] >= JOIN_MIN_PIXELS or ctx.first_block
            direct_w1 = all(_direct_grad_ok(params[6 * l + 2]) and params[6 * l + 2].grad.is_contiguous() for l in range(L))
            if USE_SIDE_STREAM and direct_w1:
                # (every gradient goes straight into .grad: nothing to hand back to autograd, the launches can be deferred)
                _defer_to_side(dev, _weight_grads)
                _side_park(dev, gbuf, buf, *dzs, *dycs, *zs)
            else:
                _weight_grads()
                for l in range(L):
                    grads[6 * l + 2] = gw1s[l]
            if STAMPS:
                stamp(f"bwd block {buf.shape[2]}x{buf.shape[3]} end (main)")
            if joins:
                _side_join(dev)
                stamp(f"bwd block {buf.shape[2]}x{buf.shape[3]} after join (main)")
            return (gbuf[:, :C0], None, *grads)
        kacc = None             # single-pass BatchNorm-1 backward: the previous pass's mean terms, [C_total][2]
        pair_a = None           # paired dx passes (Gram path): layer A's (dz, w1, gamma, beta, coef) waiting for layer B
        pair_cin = 0

        def flush_pair():
            """Layer A's term on the channels below its window, when no layer B follows on the paired path."""
            nonlocal pair_a
            if pair_a is not None:
                dzA, wA, gA, bA, cA = pair_a
                dense_bn1_dx(dzA, wA, buf[:, :pair_cin], gA, bA, stats.mean[:pair_cin], stats.rstd[:pair_cin], cA,
                             gbuf[:, :pair_cin], window=(0, pair_cin - growth))
                pair_a = None

        def gram_dx(l, dz, w1c, g1, b1, cin, coef):
            """The dx pass(es) of a Gram-path layer: alone, as layer A of a pair (windowed) or as layer B (paired).  Both
            backward schedules (side-stream lanes / serial) go through here, so they launch identical kernels."""
            nonlocal pair_a, pair_cin
            mine = (dz, w1c, g1, b1, coef)
            if pair_a is not None:
                # layer B of a pair: both layers' terms on the channels this layer reads, in one pass
                dense_bn1_dx_pair(pair_a, mine, buf[:, :cin], stats.mean[:cin], stats.rstd[:cin], gbuf[:, :cin])
                pair_a = None
            elif USE_BN1_PAIR and l >= 1 and growth == 32 and cin - growth >= 8:
                # layer A of a pair: only the 32 channels the layer below produced (its 3x3 backward reads them next);
                # the rest waits for that layer's pass
                dense_bn1_dx(dz, w1c, buf[:, :cin], g1, b1, stats.mean[:cin], stats.rstd[:cin], coef, gbuf[:, :cin],
                             window=(cin - growth, growth))
                pair_a, pair_cin = mine, cin
            else:
                dense_bn1_dx(dz, w1c, buf[:, :cin], g1, b1, stats.mean[:cin], stats.rstd[:cin], coef, gbuf[:, :cin])
        for l in range(L - 1, -1, -1):
            g1, b1, w1, g2, b2, w2 = params[6 * l: 6 * l + 6]
            a, z, a2 = saved[3 * l: 3 * l + 3]
            w1c, w2c = wcast[2 * l: 2 * l + 2]
            cin = C0 + l * growth
            m2, v2, r2 = bn2_stats[l]
            d2 = DIRECT_PARAM_GRADS and _direct_grad_ok(g2) and _direct_grad_ok(b2)
            dy_view = gbuf[:, cin:cin + growth]
            dw2_done = False
            fused2 = a2.numel() == 0 and _fused_3x3_ok(z, w2c)
            fused1 = a.numel() == 0
            main = torch.cuda.current_stream()
            side = _side_stream(z.device) if (USE_SIDE_STREAM and fused1 and fused2) else None
            if side is not None and _wrw3_direct_ok(w2, z, dy_view):
                # Main chain first, ONE fork per layer, ONE join per block.  Every cross-stream edge of the captured graph
                # costs the waiting side ~16 us (tools/trace_gaps.py): the per-layer fork + join of the first version
                # left the GPU idle for 2.4 ms/step.  Here the critical chain (conv3x3_bwd -> bn2_dz -> bn1_bwd) never
                # waits: the two atomics-bound weight-gradient kernels of the layer start on the side stream once dz
                # exists (event) and are joined only at the end of the block; dz stays referenced until then.
                d1 = DIRECT_PARAM_GRADS and _direct_grad_ok(g1) and _direct_grad_ok(b1)
                fused_wrw = _bn1_wrw_ok(w1) and z.shape[0] * z.shape[2] * z.shape[3] >= FUSED_BN1_WRW_MIN_PIXELS
                single = not fused_wrw and USE_BN1_SINGLE_PASS and z.shape[2] * z.shape[3] <= BN1_SINGLE_PASS_MAX_MAP
                fold = None
                if kacc is not None:
                    if single:      # this layer's 32 output channels: the mean terms of layer l+1, which no later pass covers
                        if FOLD_BN1_FIX and growth == 32 and _c3_flat_kernel(z.shape[3]):
                            c1_ = cin + growth          # applied inside the 3x3 backward-data kernel's dy staging
                            fold = (buf[:, cin:c1_], stats.mean[cin:c1_], stats.rstd[cin:c1_], kacc[cin:c1_])
                        else:
                            dense_bn1_fix(buf, gbuf, cin, growth, stats.mean, stats.rstd, kacc)
                    else:           # (a two-pass layer after single-pass ones: it will not apply them -- all channels now)
                        dense_bn1_fix(buf, gbuf, 0, cin + growth, stats.mean, stats.rstd, kacc)
                        kacc = None
                dz, dg2, db2, dyc = dense_conv3x3_bwd(dy_view, w2c, z, g2, b2, m2, r2, into_param_grads=d2, fix=fold)
                dy_w = dyc if dyc is not None else dy_view      # what the 3x3 weight gradient reads
                if ctx.cap is not None:
                    ctx.cap["dz"][l] = dz
                    ctx.cap.setdefault("dyc", [None] * L)[l] = dyc
                ev = torch.cuda.Event()
                ev.record(main)
                if fused_wrw:
                    # the bottleneck weight gradient rides on the BatchNorm-backward reduction (one pass over dz, x;
                    # no atomics): it is part of the main chain now, only the 3x3 weight gradient forks off
                    dg1, db1, coef = dense_bn1_wrw(dz, w1c, buf[:, :cin], g1, b1, stats.mean[:cin], stats.rstd[:cin], w1,
                                                   into_param_grads=d1)
                    gram_dx(l, dz, w1c, g1, b1, cin, coef)
                    gw1 = None
                elif single:
                    flush_pair()
                    have_prev = kacc is not None
                    if not have_prev:
                        kacc = torch.empty((buf.shape[1], 2), device=buf.device, dtype=torch.float32)
                    dg1, db1 = dense_bn1_dx_sums(dz, w1c, buf[:, :cin], g1, b1, stats.mean[:cin], stats.rstd[:cin],
                                                 gbuf[:, :cin], kacc, have_prev, into_param_grads=d1)
                else:
                    flush_pair()
                    dg1, db1 = dense_bn1_bwd(dz, w1c, buf[:, :cin], g1, b1, stats.mean[:cin], stats.rstd[:cin],
                                             gbuf[:, :cin], into_param_grads=d1)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    _run_side_pending(z.device)         # (deferred weight gradients of the block before: persistent backward)
                    dense_conv3x3_wrw(dy_w, z, g2, b2, m2, r2, w2)
                    if not fused_wrw:
                        gw1 = conv1x1_wrw(dz, buf[:, :cin], w1, bn=(g1, b1, stats.mean[:cin], stats.rstd[:cin]))
                _side_park(z.device, dz, gbuf, z, buf, dy_w)
                grads[6 * l: 6 * l + 6] = [dg1, db1, gw1, dg2, db2, None]
                continue
            gram_here = fused1 and _bn1_wrw_ok(w1) and z.shape[0] * z.shape[2] * z.shape[3] >= FUSED_BN1_WRW_MIN_PIXELS
            if not gram_here:
                flush_pair()        # (a pending layer A: its term on the lower channels, before any other kind of pass)
            main = torch.cuda.current_stream()
            side = _side_stream(z.device) if (USE_SIDE_STREAM and fused1 and fused2) else None
            sp_here = (USE_BN1_SINGLE_PASS and fused1 and z.shape[2] * z.shape[3] <= BN1_SINGLE_PASS_MAX_MAP
                       and not gram_here)
            fold = None
            if kacc is not None:
                if sp_here:                                     # the previous pass's mean terms: this layer's 32 output channels
                    if (FOLD_BN1_FIX and growth == 32 and fused2 and _c3_flat_kernel(z.shape[3])
                            and _wrw3_direct_ok(w2, z, dy_view)):
                        c1_ = cin + growth
                        fold = (buf[:, cin:c1_], stats.mean[cin:c1_], stats.rstd[cin:c1_], kacc[cin:c1_])
                    else:
                        dense_bn1_fix(buf, gbuf, cin, growth, stats.mean, stats.rstd, kacc)
                else:                                           # a two-pass layer follows: it will not apply them -- all channels now
                    dense_bn1_fix(buf, gbuf, 0, cin + growth, stats.mean, stats.rstd, kacc)
                    kacc = None
            if fold is not None:
                # (the same two kernels as the side-stream schedule: backward-data with the folded correction first, then the
                # weight gradient on the corrected copy)
                dz, dg2, db2, dyc = dense_conv3x3_bwd(dy_view, w2c, z, g2, b2, m2, r2, into_param_grads=d2, fix=fold)
                dw2_done = dense_conv3x3_wrw(dyc, z, g2, b2, m2, r2, w2)
                assert dw2_done
                dw2 = None
            elif fused2:
                # fused forward: a2 = relu(bn2(z)) was never stored.  Both kernels read dy in place from the gradient
                # buffer (row stride C_total): no contiguous copy, no MIOpen call
                if side is not None:
                    side.wait_stream(main)                      # this layer's slice of gbuf is final
                    with torch.cuda.stream(side):
                        dw2_done = dense_conv3x3_wrw(dy_view, z, g2, b2, m2, r2, w2)
                else:
                    dw2_done = dense_conv3x3_wrw(dy_view, z, g2, b2, m2, r2, w2)
            if fold is not None:
                pass
            elif dw2_done:
                dz, dg2, db2, _ = dense_conv3x3_bwd(dy_view, w2c, z, g2, b2, m2, r2, into_param_grads=d2)
                dw2 = None
            else:
                dy = dy_view                                  # (read in place through its row stride)
                if a2.numel() == 0:
                    a2 = torch.empty_like(z, memory_format=CL)
                    bn_act_fwd(z, g2, b2, m2, r2, True, a2)
                da2, dw2 = _conv_bwd(dy, a2, w2c, w2, 1, ctx.kept_cols.pop(l, None))
                dz = torch.empty_like(z, memory_format=CL)
                dg2, db2 = bn_act_bwd(dense_cl(da2), z, g2, b2, m2, r2, True, dz, False,
                                      into_param_grads=d2)
            d1 = DIRECT_PARAM_GRADS and _direct_grad_ok(g1) and _direct_grad_ok(b1)
            if fused1:
                # fused forward: nothing of norm1's output was kept.  Weight gradient with BN1+ReLU recomputed from the
                # concat buffer; data gradient + BN1 backward without materialising da
                bn1 = (g1, b1, stats.mean[:cin], stats.rstd[:cin])
                if gram_here:
                    dw1 = ("direct", None)
                    dg1, db1, coef = dense_bn1_wrw(dz, w1c, buf[:, :cin], g1, b1, stats.mean[:cin], stats.rstd[:cin], w1,
                                                   into_param_grads=d1)
                    gram_dx(l, dz, w1c, g1, b1, cin, coef)
                    if side is not None:
                        main.wait_stream(side)
                    grads[6 * l: 6 * l + 6] = [dg1, db1, None, dg2, db2, None if (dw2_done or dw2 is None) else _wgrad(w2, dw2)]
                    continue
                if side is not None and dw2_done:
                    side.wait_stream(main)                      # dz is ready
                    with torch.cuda.stream(side):
                        dw1 = ("direct", conv1x1_wrw(dz, buf[:, :cin], w1, bn=bn1))
                else:
                    dw1 = ("direct", conv1x1_wrw(dz, buf[:, :cin], w1, bn=bn1))
                if sp_here:
                    have_prev = kacc is not None                # (the same kernels as the side-stream schedule above)
                    if not have_prev:
                        kacc = torch.empty((buf.shape[1], 2), device=buf.device, dtype=torch.float32)
                    dg1, db1 = dense_bn1_dx_sums(dz, w1c, buf[:, :cin], g1, b1, stats.mean[:cin], stats.rstd[:cin],
                                                 gbuf[:, :cin], kacc, have_prev, into_param_grads=d1)
                else:
                    dg1, db1 = dense_bn1_bwd(dz, w1c, buf[:, :cin], g1, b1, stats.mean[:cin], stats.rstd[:cin],
                                             gbuf[:, :cin], into_param_grads=d1)
                if side is not None:
                    main.wait_stream(side)                      # join: dz / dy may be released or overwritten now
            else:
                da, dw1 = _conv1x1_bwd(dz, a, w1c, w1)
                dg1, db1 = bn_act_bwd(da, buf[:, :cin], g1, b1, stats.mean[:cin],
                                      stats.rstd[:cin], True, gbuf[:, :cin], True, into_param_grads=d1)
            gw1 = dw1[1] if isinstance(dw1, tuple) else (None if dw1 is None else _wgrad(w1, dw1))
            grads[6 * l: 6 * l + 6] = [dg1, db1, gw1, dg2, db2, None if (dw2_done or dw2 is None) else _wgrad(w2, dw2)]
        # The tensors the side stream still reads stay parked (referenced) until a join.  Joining after every block makes
        # the main chain wait whenever the side stream runs behind; the small maps can afford to keep their tensors alive
        # (tens of MB) until a later block joins: 14.58 -> 14.28 ms/step on configs[1].  The network's first block is the
        # last one of the backward: it always joins, so nothing is left running when the backward returns.
        flush_pair()
        if kacc is not None:
            dense_bn1_fix(buf, gbuf, 0, C0, stats.mean, stats.rstd, kacc)     # the block input: the mean terms of layer 0
        if STAMPS:
            stamp(f"bwd block {buf.shape[2]}x{buf.shape[3]} end (main)")
            if USE_SIDE_STREAM:
                with torch.cuda.stream(_side_stream(buf.device)):
                    stamp(f"bwd block {buf.shape[2]}x{buf.shape[3]} weight gradients done (side)")
        if buf.shape[0] * buf.shape[2] * buf.shape[3] >= JOIN_MIN_PIXELS or ctx.first_block:
            _side_join(buf.device)
            stamp(f"bwd block {buf.shape[2]}x{buf.shape[3]} after join (main)")
        return (gbuf[:, :C0], None, *grads)


def _bn_train(x: Tensor, bn: nn.BatchNorm2d, relu: bool, rec: _RunningStats) -> Tensor:
    C = x.shape[1]
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    var = torch.empty_like(mean)
    rstd = torch.empty_like(mean)
    x = dense_cl(x)
    bn_stats(x, mean, var, rstd, bn.eps)
    rec.add(bn, mean, var, x.numel() // C)
    return BNActFn.apply(x, bn.weight, bn.bias, mean, rstd, relu)


def dense_block(blk: nn.Module, x: Tensor, rec: _RunningStats, stats: Optional[_BlockStats] = None,
                force_join: bool = False) -> Tuple[Tensor, _BlockStats]:
    """``stats``: a _BlockStats whose first C0 entries already hold the statistics of ``x`` (produced by the
    transition's convolution epilogue) -- the block then only copies x into its buffer."""
    layers = list(blk.values())
    growth = layers[0].conv2.out_channels
    C0 = x.shape[1]
    Ct = C0 + len(layers) * growth
    prefilled = stats is not None
    if stats is None:
        stats = _BlockStats(Ct, x.device)
    bott = layers[0].conv1.out_channels
    bn2 = [tuple(torch.empty(bott, device=x.device, dtype=torch.float32) for _ in range(3)) for _ in layers]
    params = []
    for ly in layers:
        params += [ly.norm1.weight, ly.norm1.bias, ly.conv1.weight, ly.norm2.weight, ly.norm2.bias, ly.conv2.weight]
    meta = (stats, [ly.norm1.eps for ly in layers], [ly.norm2.eps for ly in layers], growth, bn2, prefilled, force_join)
    buf = DenseBlockFn.apply(x, meta, *params)
    n = x.shape[0] * x.shape[2] * x.shape[3]
    for i, ly in enumerate(layers):
        cin = C0 + i * growth
        rec.add(ly.norm1, stats.mean[:cin], stats.var[:cin], n)
        rec.add(ly.norm2, bn2[i][0], bn2[i][1], n)
    return buf, stats


# --------------------------------------------------------------------------- stem convolution (csrc/dense_conv.hip)


def _conv0_ok(x: Tensor, conv: nn.Conv2d) -> bool:
    B, C, H, W = x.shape
    return (x.is_cuda and x.dtype == torch.bfloat16 and C == 3 and H % 4 == 0 and W % 8 == 0
            and W <= 256 and x.is_contiguous(memory_format=CL) and tuple(conv.weight.shape) == (64, 3, 7, 7)
            and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.bias is None and not x.requires_grad)


def conv0_fwd(x: Tensor, w16: Tensor, eps: float, stats: Optional[Tuple[Tensor, Tensor, Tensor]]) -> Tensor:
    """y = conv2d(x, w16, stride 2, padding 3) for the 7x7 stem convolution, bf16 NHWC, and (``stats`` = (mean, var,
    rstd) tensors of 64 floats, or None) the batch statistics of y for norm0 from the kernel's epilogue."""
    B, _, H, W = x.shape
    y = torch.empty((B, 64, H // 2, W // 2), device=x.device, dtype=torch.bfloat16, memory_format=CL)
    L = _lib.lib()
    ws = _ws(L.mcl_conv0_workspace_floats(B, H, W), x.device)
    st = (None, None, None) if stats is None else tuple(t.data_ptr() for t in stats)
    check(L.mcl_conv0_fwd(x.data_ptr(), B, H, W, w16.data_ptr(), y.data_ptr(), ws.data_ptr(), eps, *st, _stream()),
          "mcl_conv0_fwd")
    return y


class Conv0Fn(torch.autograd.Function):
    """The stem convolution on the hand-written kernel; meta = (eps, (mean, var, rstd)) receives norm0's batch
    statistics.  Backward: weight gradient only (the image does not require a gradient)."""

    @staticmethod
    def forward(ctx, x, w, meta):
        eps, stats = meta
        w16 = _weight(w, x.dtype)
        y = conv0_fwd(x, w16, eps, stats)
        ctx.save_for_backward(x, w16)
        ctx.w = w
        ctx.cap = None
        if CAPTURE_MISC is not None:
            ctx.cap = {"kind": "conv0", "x": x, "w16": w16, "y": y, "stats": stats, "w": w}
            CAPTURE_MISC.append(ctx.cap)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w16 = ctx.saved_tensors
        dy = dense_cl(dy)
        w = ctx.w
        if ctx.cap is not None:
            ctx.cap["dy"] = dy
        B, _, H, W = x.shape
        if DIRECT_PARAM_GRADS and _direct_grad_ok(w) and dy.dtype == torch.bfloat16:
            L = _lib.lib()
            ws = _ws(L.mcl_conv0_wrw_workspace_floats(B, H, W), x.device)
            tgt, in_place = _grad_target_khwc(w)
            check(L.mcl_conv0_wrw(x.data_ptr(), B, H, W, dy.data_ptr(), ws.data_ptr(), tgt.data_ptr(), int(in_place),
                                  _stream()), "mcl_conv0_wrw")                          # straight into the fp32 .grad
            _grad_finish_khwc(w, tgt, in_place)
            return None, None, None
        from . import conv_generic as cg
        dw = cg.conv_bwd_weight(dy, x, ctx.w, 2, 3)          # (parameter without a dense fp32 .grad: generic path)
        return None, (None if dw is None else dw.to(ctx.w.dtype)), None


# --------------------------------------------------------------------------- pooling (csrc/pool.hip)


def _pool_ok(x: Tensor, even: bool) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=CL) and (not even or (x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)))


class AvgPool2Fn(torch.autograd.Function):
    """nn.AvgPool2d(2, 2) of the transitions on channels-last bf16."""

    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        y = torch.empty((B, C, H // 2, W // 2), device=x.device, dtype=x.dtype, memory_format=CL)
        check(_lib.lib().mcl_avgpool2_nhwc_bf16(x.data_ptr(), y.data_ptr(), B, H, W, C, 0, _stream()), "mcl_avgpool2")
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = dense_cl(dy)
        dx = torch.empty((B, C, H, W), device=dy.device, dtype=dy.dtype, memory_format=CL)
        check(_lib.lib().mcl_avgpool2_nhwc_bf16(dy.data_ptr(), dx.data_ptr(), B, H, W, C, 1, _stream()), "mcl_avgpool2")
        return dx


class MaxPool3s2Fn(torch.autograd.Function):
    """nn.MaxPool2d(3, stride=2, padding=1) (pool0) on channels-last bf16; arg-max recomputed in the backward."""

    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        y = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=x.device, dtype=x.dtype, memory_format=CL)
        idx = torch.empty((B, y.shape[2], y.shape[3], C), device=x.device, dtype=torch.uint8)
        check(_lib.lib().mcl_maxpool3s2_nhwc_bf16_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), B, H, W, C, _stream()),
              "mcl_maxpool")
        ctx.save_for_backward(idx)
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, C, H, W = ctx.shape
        dy = dense_cl(dy)
        dx = torch.empty((B, C, H, W), device=dy.device, dtype=dy.dtype, memory_format=CL)
        check(_lib.lib().mcl_maxpool3s2_nhwc_bf16_bwd(idx.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, H, W, C, _stream()),
              "mcl_maxpool bwd")
        return dx


class StemTailFn(torch.autograd.Function):
    """norm0 -> relu0 -> pool0 of the DenseNet stem as one pass over the conv0 output (csrc/pool.hip, csrc/bnrelu.hip):
    the normalised full-resolution map is never written (the backward recomputes the ReLU mask from x)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, rstd):
        B, C, H, W = x.shape
        y = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=x.device, dtype=x.dtype, memory_format=CL)
        idx = torch.empty((B, y.shape[2], y.shape[3], C), device=x.device, dtype=torch.uint8)
        check(_lib.lib().mcl_bn_act_maxpool_fwd(x.data_ptr(), B, H, W, C, gamma.data_ptr(), beta.data_ptr(),
                                                mean.data_ptr(), rstd.data_ptr(), y.data_ptr(), idx.data_ptr(),
                                                _stream()), "mcl_bn_act_maxpool_fwd")
        ctx.save_for_backward(x, idx, mean, rstd)
        ctx.params = (gamma, beta)
        ctx.cap = None
        if CAPTURE_MISC is not None:
            ctx.cap = {"kind": "stem_tail", "x": x, "y": y, "mean": mean, "rstd": rstd, "params": (gamma, beta)}
            CAPTURE_MISC.append(ctx.cap)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, idx, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        B, C, H, W = x.shape
        # dy = the channel slice [:C] of the first dense block's gradient buffer: read in place through its row stride
        pdy, _, Cd, lddy = _rows(dy)
        assert Cd == C and dy.dtype == x.dtype
        if ctx.cap is not None:
            ctx.cap["dy"] = dy.clone(memory_format=CL)
        g = torch.empty_like(x, memory_format=CL)           # un-pooled gradient (gather, deterministic)
        check(_lib.lib().mcl_maxpool3s2_nhwc_bf16_bwd_ld(idx.data_ptr(), pdy, lddy, g.data_ptr(), B, H, W, C, _stream()),
              "mcl_maxpool bwd")
        direct = DIRECT_PARAM_GRADS and _direct_grad_ok(gamma) and _direct_grad_ok(beta)
        dg, db = bn_act_bwd(g, x, gamma, beta, mean, rstd, True, g, False, into_param_grads=direct)   # dx in place of g
        if ctx.cap is not None:
            ctx.cap["dx"] = g
        return g, dg, db, None, None




def _stem_tail_ok(x: Tensor) -> bool:
    return _pool_ok(x, False) and x.shape[1] <= 2048


def max_pool_3s2(x: Tensor) -> Tensor:
    if _pool_ok(x, False):
        return MaxPool3s2Fn.apply(x)
    from . import conv_generic as cg
    return cg.max_pool_3s2(x)                       # fp32 activations (csrc/pool_generic.hip)


def avg_pool_2(x: Tensor) -> Tensor:
    if _pool_ok(x, True):
        return AvgPool2Fn.apply(x)
    from . import conv_generic as cg
    return cg.avg_pool_2(x)                         # fp32 activations / odd maps


# --------------------------------------------------------------------------- transitions
# torchvision _Transition = norm -> relu -> conv1x1 (C -> C/2) -> AvgPool2d(2, 2).  The convolution is linear and per
# pixel, so it commutes with the pool: p = avgpool(relu(bn(buf))) is formed first (csrc/bnrelu.hip) and the
# convolution, its weight gradient and its data gradient all run on a QUARTER of the pixels, on the same kernels as
# the dense layers' bottleneck convolution (identity BN prologue: p >= 0 so relu(1*p + 0) == p); its epilogue yields
# the batch statistics the next block's norm1 layers need, so that pass disappears too.
_ident_cache = {}


def _identity_bn(device) -> Tuple[Tensor, Tensor]:
    v = _ident_cache.get(device.index)
    if v is None:
        v = (torch.ones(1024, device=device, dtype=torch.float32), torch.zeros(1024, device=device, dtype=torch.float32))
        _ident_cache[device.index] = v
    return v


def bn_act_avgpool_fwd(x: Tensor, gamma: Tensor, beta: Tensor, mean: Tensor, rstd: Tensor) -> Tensor:
    B, C, H, W = x.shape
    px, S, _, ld = _rows(x)
    p = torch.empty((B, C, H // 2, W // 2), device=x.device, dtype=x.dtype, memory_format=CL)
    check(_lib.lib().mcl_bn_act_avgpool_fwd(px, ld, B, H, W, C, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                            rstd.data_ptr(), p.data_ptr(), C, _stream()), "mcl_bn_act_avgpool_fwd")
    return p


def pooled_conv1x1_fwd(p: Tensor, w16: Tensor, eps: float, stats: Optional[_BlockStats], out: Optional[Tensor] = None
                       ) -> Tensor:
    """y = conv1x1(p, w16) for C_out a multiple of 128 (one launch per 128 output channels, each with its statistics
    epilogue writing stats.mean/var/rstd[n0:n0+128]); ``stats`` None: no statistics (inference).  ``out``: a channels-last
    (possibly channel-sliced) view that receives y -- the first C_out channels of the next dense block's concat buffer."""
    B, C, H, W = p.shape
    Co = w16.shape[0]
    y = out if out is not None else torch.empty((B, Co, H, W), device=p.device, dtype=p.dtype, memory_format=CL)
    py, Sy, Cy, ldy = _rows(y)
    one, zero = _identity_bn(p.device)
    pp, S, _, ldp = _rows(p)
    assert (Sy, Cy) == (S, Co) and y.dtype == p.dtype
    L = _lib.lib()
    ws = _ws(L.mcl_dense_conv1x1_workspace_floats(S), p.device)
    for n0 in range(0, Co, 128):
        st = (None, None, None) if stats is None else tuple(t.data_ptr() + 4 * n0 for t in (stats.mean, stats.var, stats.rstd))
        check(L.mcl_dense_conv1x1_fwd(pp, ldp, S, C, one.data_ptr(), zero.data_ptr(), zero.data_ptr(), one.data_ptr(),
                                      w16.data_ptr() + 2 * n0 * C, py + 2 * n0, ldy, ws.data_ptr(), eps,
                                      *st, _stream()), "mcl_dense_conv1x1_fwd (transition)")
    return y


def _transition_ok(buf: Tensor, w: Tensor) -> bool:
    B, C, H, W = buf.shape
    # odd maps (her2st: 112 px patches reach 7 x 7 at the last transition) pool with floor, like nn.AvgPool2d(2, 2)
    return (buf.is_cuda and buf.dtype == torch.bfloat16 and H >= 2
            and W >= 2 and C % 8 == 0 and C <= 1024 and w.shape[0] % 128 == 0 and w.shape[1] == C
            and buf.is_contiguous(memory_format=CL))


class TransitionFn(torch.autograd.Function):
    """(B, C, H, W) block buffer -> (B, C/2, H/2, W/2); meta = (stats of the buffer, _BlockStats of the NEXT block
    whose first C/2 entries are filled here, eps of the next block's norm1)."""

    @staticmethod
    def forward(ctx, buf, gamma, beta, w, meta):
        stats, next_stats, eps_next = meta
        p = bn_act_avgpool_fwd(buf, gamma, beta, stats.mean, stats.rstd)
        w16 = _weight(w, buf.dtype)
        nbuf = getattr(next_stats, "buf", None)
        out = nbuf[:, :w16.shape[0]] if nbuf is not None else None
        y = pooled_conv1x1_fwd(p, w16, eps_next, next_stats, out=out)
        ctx.save_for_backward(buf, p, w16)
        ctx.params = (gamma, beta, w)
        ctx.stats = stats
        ctx.cap = None
        if CAPTURE_MISC is not None:
            ctx.cap = {"kind": "transition", "buf": buf, "p": p, "w16": w16, "y": y, "stats": stats, "next_stats": next_stats,
                       "params": (gamma, beta, w), "eps_next": eps_next}
            CAPTURE_MISC.append(ctx.cap)
        return y

    @staticmethod
    def backward(ctx, dy):
        buf, p, w16 = ctx.saved_tensors
        gamma, beta, w = ctx.params
        stats = ctx.stats
        B, C, H, W = buf.shape
        Co = w16.shape[0]
        _rows(dy)                                             # channels-last (possibly channel-sliced) view
        main = torch.cuda.current_stream()
        deferred = (USE_SIDE_STREAM and DIRECT_PARAM_GRADS and _direct_grad_ok(w)
                    and w.grad.is_contiguous())
        if deferred:
            ev = torch.cuda.Event()
            ev.record(main)                                   # dy is final
        # dp = dy . W: (S/4, Co) x (Co, C), the weight consumed in place as the reduction-major operand of mcl_gemm_bf16
        # (csrc/gemm_bf16.hip); dy may be the channel slice [:Co] of the next block's gradient buffer (row stride lddy)
        pdy, Sq, Co_, lddy = _rows(dy)
        assert Co_ == Co and dy.dtype == torch.bfloat16
        dp = torch.empty((B, C, dy.shape[2], dy.shape[3]), device=buf.device, dtype=torch.bfloat16, memory_format=CL)
        check(_lib.lib().mcl_gemm_bf16(pdy, lddy, 0, w16.data_ptr(), C, 0, dp.data_ptr(), C, 0, Sq, C, Co, 1, 1, 0, 0, 0,
                                       1.0, 2, None, None, 0, 0, None, 0, None, 0, 1, None, 0, _stream()),
              "mcl_gemm_bf16 (transition backward-data)")
        dx = torch.empty_like(buf, memory_format=CL)
        direct = DIRECT_PARAM_GRADS and _direct_grad_ok(gamma) and _direct_grad_ok(beta)
        if direct:
            dg, db = gamma.grad, beta.grad
        else:
            dg = torch.empty(C, device=buf.device, dtype=torch.float32)
            db = torch.empty(C, device=buf.device, dtype=torch.float32)
        L = _lib.lib()
        ws = _ws(L.mcl_bn_workspace_floats(B * H * W, C, 1), buf.device)
        check(L.mcl_bn_act_avgpool_bwd(dp.data_ptr(), C, buf.data_ptr(), C, B, H, W, C, gamma.data_ptr(),
                                       beta.data_ptr(), stats.mean.data_ptr(), stats.rstd.data_ptr(), ws.data_ptr(),
                                       dg.data_ptr(), db.data_ptr(), int(direct), dx.data_ptr(), C, _stream()),
              "mcl_bn_act_avgpool_bwd")
        if deferred:
            # weight gradient (atomics-bound, needs only dy and p) on the side stream; joined by the dense block that
            # follows in the backward order
            side = _side_stream(buf.device)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                _run_side_pending(buf.device)            # (deferred weight gradients of the dense block before)
                dw = conv1x1_wrw(dy, p, w)
            _side_park(buf.device, dy, p)
        else:
            dw = conv1x1_wrw(dy, p, w)                        # dW += dy^T p, straight into w.grad when it exists
        gw = None if dw is None else dw.view_as(w).to(w.dtype)
        if ctx.cap is not None:
            ctx.cap.update({"dy": dy, "dp": dp, "dx": dx.clone(memory_format=CL)})   # (dx becomes a gradient buffer, in place)
        return dx, (None if direct else dg), (None if direct else db), gw, None


class BNGlobalPoolFn(torch.autograd.Function):
    """norm5 -> F.adaptive_avg_pool2d(., (1, 1)) -> flatten, the tail of /root/reference/model.py:81-85 (no ReLU between
    them), as ONE forward launch and two backward launches (csrc/step_misc.hip): the pool of an affine map is the affine
    map of the pool, so the normalised map is never written, and the BatchNorm backward of a gradient that is constant
    over each image's map needs only sums over the B images.  (B, C, H, W) bf16 channels-last -> (B, C) fp32."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, rstd):
        B, C, H, W = x.shape
        px, S, _, ld = _rows(x)
        out = torch.empty((B, C), device=x.device, dtype=torch.float32)
        xm = torch.empty((B, C), device=x.device, dtype=torch.float32)
        check(_lib.lib().mcl_bn_gap_fwd(px, ld, B, H * W, C, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                        rstd.data_ptr(), out.data_ptr(), xm.data_ptr(), _stream()), "mcl_bn_gap_fwd")
        ctx.save_for_backward(x, xm, mean, rstd)
        ctx.params = (gamma, beta)
        ctx.cap = None
        if CAPTURE_MISC is not None:
            ctx.cap = {"kind": "norm5_pool", "x": x, "out": out, "mean": mean, "rstd": rstd, "params": (gamma, beta)}
            CAPTURE_MISC.append(ctx.cap)
        return out

    @staticmethod
    def backward(ctx, g):
        x, xm, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        B, C, H, W = x.shape
        g = g.contiguous()
        if g.dtype != torch.float32:
            g = g.float()
        px, S, _, ld = _rows(x)
        dx = torch.empty_like(x, memory_format=CL)
        direct = DIRECT_PARAM_GRADS and _direct_grad_ok(gamma) and _direct_grad_ok(beta)
        if direct:
            dg, db = gamma.grad, beta.grad
        else:
            dg = torch.empty(C, device=x.device, dtype=torch.float32)
            db = torch.empty(C, device=x.device, dtype=torch.float32)
        coef = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        check(_lib.lib().mcl_bn_gap_bwd(g.data_ptr(), xm.data_ptr(), px, ld, B, H * W, C, gamma.data_ptr(), mean.data_ptr(),
                                        rstd.data_ptr(), coef.data_ptr(), dg.data_ptr(), db.data_ptr(), int(direct),
                                        dx.data_ptr(), C, _stream()), "mcl_bn_gap_bwd")
        if ctx.cap is not None:
            ctx.cap.update({"g": g, "dx": dx.clone(memory_format=CL)})     # (dx becomes block 4's gradient buffer, in place)
        return dx, (None if direct else dg), (None if direct else db), None, None


def _gap_ok(buf: Tensor) -> bool:
    return (buf.is_cuda and buf.dtype == torch.bfloat16 and buf.dim() == 4 and buf.shape[1] % 8 == 0
            and buf.is_contiguous(memory_format=CL))


def image_to_act(x: Tensor, act_dtype: torch.dtype, out: Optional[Tensor] = None) -> Tensor:
    """``x.to(act_dtype).contiguous(memory_format=channels_last)`` of the input image; fp32 -> bf16 as one launch of
    mcl_image_to_bf16_nhwc for any input strides (NCHW from a DataLoader, channels-last from bench.py).  ``out``: a bf16
    channels-last tensor that receives the result (engine.TrainStep's static graph input: the per-step input copy IS the cast)."""
    if (act_dtype == torch.bfloat16 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not x.requires_grad
            and x.numel() > 0):
        B, C, H, W = x.shape
        if out is not None:
            assert out.shape == x.shape and out.dtype == torch.bfloat16 and out.is_contiguous(memory_format=CL)
            y = out
        else:
            y = torch.empty((B, C, H, W), device=x.device, dtype=torch.bfloat16, memory_format=CL)
        sb, sc, sy, sx = x.stride()
        check(_lib.lib().mcl_image_to_bf16_nhwc(x.data_ptr(), sb, sc, sy, sx, B, C, H, W, y.data_ptr(), _stream()),
              "mcl_image_to_bf16_nhwc")
        return y
    y = x.to(dtype=act_dtype).contiguous(memory_format=CL)
    if out is not None:
        out.copy_(y)
        return out
    return y


# ---- backward in SEGMENTS (data parallel: engine.TrainStep captures one HIP graph per segment and issues the all-reduce of
# a segment's gradient range while the next segment replays).  A cut sits at every dense block's input behind a transition:
# the forward passes the tensor through unchanged; the backward stops there, leaving the incoming gradient in ``slot["grad"]``;
# the next segment is ``torch.autograd.backward((upstream,), (slot["grad"],))``.
_cut_anchors = {}


def _cut_anchor(device) -> Tensor:
    a = _cut_anchors.get(device.index)
    if a is None:
        a = torch.zeros(1, device=device, requires_grad=True)     # makes the cut's output require grad; never receives one
        _cut_anchors[device.index] = a
    return a


class _CutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, slot):
        ctx.slot = slot
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.slot["grad"] = g
        return None, None, None


# dense block index -> callable invoked on the main stream right after that block's forward has been enqueued
FORWARD_BLOCK_HOOKS: dict = {}


def densenet_features_fused(features: nn.Sequential, x: Tensor, act_dtype: torch.dtype = torch.bfloat16,
                            pooled: bool = False, cuts: Optional[list] = None, cut_blocks=()) -> Tensor:
    """Train-mode forward of torchvision-layout DenseNet ``features`` (conv0 ... norm5), returning the
    (B, C, h, w) norm5 output (no ReLU: model.py:82-84 pools it directly) or, ``pooled``, the (B, C) fp32 features
    after model.py:83-84's adaptive_avg_pool2d + flatten (norm5 and the pool as one kernel on the bf16 path).
    ``cuts`` (a list) + ``cut_blocks`` (dense block indices, 2..): the backward is cut at the input of those blocks (behind
    their transition); (upstream tensor, slot) pairs are appended to ``cuts`` in forward order (see ``_CutFn``)."""
    if not x.is_cuda:
        raise RuntimeError("densenet_features_fused: input is on the CPU; the fused backbone path is GPU-only")
    rec = _RunningStats()
    x = image_to_act(x, act_dtype)
    own_conv0 = _conv0_ok(x, features.conv0)
    if own_conv0:
        mean0, var0, rstd0 = (torch.empty(64, device=x.device, dtype=torch.float32) for _ in range(3))
        x = Conv0Fn.apply(x, features.conv0.weight, (features.norm0.eps, (mean0, var0, rstd0)))
    else:
        from . import conv_generic as cg
        x = cg.conv2d(x, features.conv0.weight, features.conv0.stride[0], features.conv0.padding[0])
    x = dense_cl(x)
    if _stem_tail_ok(x):
        C0 = x.shape[1]
        if not own_conv0:
            mean0, var0, rstd0 = (torch.empty(C0, device=x.device, dtype=torch.float32) for _ in range(3))
            bn_stats(x, mean0, var0, rstd0, features.norm0.eps)
        rec.add(features.norm0, mean0, var0, x.numel() // C0)
        x = StemTailFn.apply(x, features.norm0.weight, features.norm0.bias, mean0, rstd0)
    else:
        x = _bn_train(x, features.norm0, True, rec)
        x = max_pool_3s2(dense_cl(x))
    i = 1
    out = None
    next_stats = None
    cut_here = False
    while hasattr(features, f"denseblock{i}"):
        buf, stats = dense_block(getattr(features, f"denseblock{i}"), x, rec, next_stats, force_join=cut_here)
        stamp(f"fwd block {i} done (main)")
        hook = FORWARD_BLOCK_HOOKS.get(i)
        if hook is not None:
            hook()                                            # e.g. an event other streams wait for (model.embed)
        next_stats = None
        n = buf.shape[0] * buf.shape[2] * buf.shape[3]
        if hasattr(features, f"transition{i}"):
            tr = getattr(features, f"transition{i}")
            rec.add(tr.norm, stats.mean, stats.var, n)
            nxt = list(getattr(features, f"denseblock{i + 1}").values()) if hasattr(features, f"denseblock{i + 1}") else None
            if nxt is not None and _transition_ok(buf, tr.conv.weight):
                co = tr.conv.weight.shape[0]
                next_stats = _BlockStats(co + len(nxt) * nxt[0].conv2.out_channels, buf.device)
                # the next block's concat buffer exists already: the transition's convolution writes its first `co` channels
                next_stats.buf = torch.empty((buf.shape[0], next_stats.mean.numel(), buf.shape[2] // 2, buf.shape[3] // 2),
                                             device=buf.device, dtype=buf.dtype, memory_format=CL)
                x = TransitionFn.apply(buf, tr.norm.weight, tr.norm.bias, tr.conv.weight,
                                       (stats, next_stats, nxt[0].norm1.eps))
                cut_here = cuts is not None and (i + 1) in cut_blocks and x.requires_grad
                if cut_here:
                    slot = {}
                    cuts.append((x, slot))
                    x = _CutFn.apply(x.detach(), _cut_anchor(x.device), slot)
            else:
                from . import conv_generic as cg
                a = BNActFn.apply(buf, tr.norm.weight, tr.norm.bias, stats.mean, stats.rstd, True)
                x = avg_pool_2(cg.conv2d(a, tr.conv.weight, 1, 0))
        else:
            if pooled and _gap_ok(buf):
                out = BNGlobalPoolFn.apply(buf, features.norm5.weight, features.norm5.bias, stats.mean, stats.rstd)
            else:
                out = BNActFn.apply(buf, features.norm5.weight, features.norm5.bias, stats.mean, stats.rstd, False)
                if pooled:
                    from . import conv_generic as _cg
                    out = _cg.global_avg_pool(out)        # own kernel, fp32 (B, C) (was F.adaptive_avg_pool2d)
            rec.add(features.norm5, stats.mean, stats.var, n)
        i += 1
    rec.flush()
    return out


# --------------------------------------------------------------------------- inference (eval mode, running statistics)
@torch.no_grad()
def eval_rstd(bns) -> List[Tensor]:
    """1 / sqrt(running_var + eps) of every BatchNorm layer in two launches (mcl_bn_eval_rstd: pointer table by value)."""
    import ctypes as C
    if not bns:
        return []
    dev = bns[0].running_var.device
    sizes = [bn.running_var.numel() for bn in bns]
    flat = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
    outs, o = [], 0
    for n in sizes:
        outs.append(flat[o:o + n])
        o += n
    for bn in bns:
        if bn.running_var.dtype != torch.float32 or not bn.running_var.is_contiguous():
            raise RuntimeError("eval_rstd: running statistics must be contiguous fp32")
    n = len(bns)
    vp = C.c_void_p * n
    check(_lib.lib().mcl_bn_eval_rstd(n, vp(*[bn.running_var.data_ptr() for bn in bns]), vp(*[t.data_ptr() for t in outs]),
                                      (C.c_int32 * n)(*sizes), (C.c_float * n)(*[float(bn.eps) for bn in bns]), _stream()),
          "mcl_bn_eval_rstd")
    return outs


def densenet_features_eval(features: nn.Sequential, x: Tensor, act_dtype: torch.dtype = torch.bfloat16,
                           pooled: bool = False) -> Tensor:
    """Eval-mode forward of torchvision-layout DenseNet ``features`` (what ``model.image_encoder`` computes for
    /root/reference/evel_her2st.py:50) on the same fused kernels as training: every BatchNorm is the affine map of
    its RUNNING statistics, applied inside the convolution prologues, so a dense layer is two launches (no statistics,
    no finalize) and nothing but the concat buffers and the 128-channel bottleneck outputs touches HBM."""
    if not x.is_cuda:
        raise RuntimeError("densenet_features_eval: input is on the CPU; the fused backbone path is GPU-only")
    if act_dtype not in (torch.bfloat16, torch.float32):
        raise RuntimeError("densenet_features_eval: activations are bf16 (fused kernels) or fp32 (generic own-kernel path)")
    bns = [m for m in features.modules() if isinstance(m, nn.BatchNorm2d)]
    rs = {id(bn): r for bn, r in zip(bns, eval_rstd(bns))}

    def affine(t: Tensor, bn: nn.BatchNorm2d, relu: bool) -> Tensor:
        t = dense_cl(t)
        out = torch.empty_like(t, memory_format=CL)
        bn_act_fwd(t, bn.weight, bn.bias, bn.running_mean, rs[id(bn)], relu, out)
        return out

    def copy_into(dst: Tensor, src: Tensor) -> None:
        # an identity affine on the own BatchNorm kernel: a strided copy without an ATen launch
        one, zero = _identity_bn(src.device)
        C = src.shape[1]
        bn_act_fwd(src, one[:C], zero[:C], zero[:C], one[:C], False, dst)

    x = image_to_act(x, act_dtype)
    if _conv0_ok(x, features.conv0):
        x = conv0_fwd(x, _weight(features.conv0.weight, act_dtype), features.norm0.eps, None)
    else:
        from . import conv_generic as cg
        x = cg.conv_fwd(x, _weight(features.conv0.weight, act_dtype), features.conv0.stride[0], features.conv0.padding[0])
    if not x.is_contiguous(memory_format=CL):
        raise RuntimeError("densenet_features_eval: the stem convolution must produce a channels-last tensor")
    if _stem_tail_ok(x):
        x = StemTailFn.apply(x, features.norm0.weight, features.norm0.bias, features.norm0.running_mean,
                             rs[id(features.norm0)])
    else:
        x = max_pool_3s2(affine(x, features.norm0, True))
    i = 1
    out = None
    pre = None                      # the next block's concat buffer when the transition wrote its output straight into it
    while hasattr(features, f"denseblock{i}"):
        layers = list(getattr(features, f"denseblock{i}").values())
        growth = layers[0].conv2.out_channels
        B, C0, H, W = x.shape
        if pre is not None:
            buf = pre
        else:
            buf = torch.empty((B, C0 + len(layers) * growth, H, W), device=x.device, dtype=act_dtype, memory_format=CL)
            copy_into(buf[:, :C0], x)
        pre = None
        for l, ly in enumerate(layers):
            cin = C0 + l * growth
            w1c, w2c = _weight(ly.conv1.weight, act_dtype), _weight(ly.conv2.weight, act_dtype)
            if growth == 32 and _fused_1x1_ok(buf, w1c) and w1c.shape[0] == 128:
                z = dense_conv1x1_fwd(buf[:, :cin], ly.norm1.weight, ly.norm1.bias, ly.norm1.running_mean,
                                      rs[id(ly.norm1)], w1c, ly.norm2.eps, None, None, None)
                dense_conv3x3_fwd(z, ly.norm2.weight, ly.norm2.bias, ly.norm2.running_mean, rs[id(ly.norm2)], w2c,
                                  buf[:, cin:cin + growth], ly.norm1.eps, None, None, None)
            else:
                from . import conv_generic as cg
                a = torch.empty((B, cin, H, W), device=x.device, dtype=act_dtype, memory_format=CL)
                bn_act_fwd(buf[:, :cin], ly.norm1.weight, ly.norm1.bias, ly.norm1.running_mean, rs[id(ly.norm1)], True, a)
                z = affine(_conv1x1_fwd(a, w1c), ly.norm2, True)
                cg.conv_fwd(z, w2c, 1, 1, out=buf[:, cin:cin + growth])
        if hasattr(features, f"transition{i}"):
            tr = getattr(features, f"transition{i}")
            nxt = getattr(features, f"denseblock{i + 1}", None)
            Co = tr.conv.weight.shape[0]
            Hn, Wn = buf.shape[2] // 2, buf.shape[3] // 2
            if nxt is not None:
                nl = list(nxt.values())
                pre = torch.empty((buf.shape[0], Co + len(nl) * nl[0].conv2.out_channels, Hn, Wn), device=buf.device,
                                  dtype=act_dtype, memory_format=CL)
            if _transition_ok(buf, tr.conv.weight):
                p = bn_act_avgpool_fwd(buf, tr.norm.weight, tr.norm.bias, tr.norm.running_mean, rs[id(tr.norm)])
                x = pooled_conv1x1_fwd(p, _weight(tr.conv.weight, act_dtype), tr.norm.eps, None,
                                       out=None if pre is None else pre[:, :Co])
            else:
                from . import conv_generic as cg
                a = affine(buf, tr.norm, True)
                x = avg_pool_2(cg.conv_fwd(a, _weight(tr.conv.weight, act_dtype), 1, 0))
                if pre is not None:
                    copy_into(pre[:, :Co], x)
                    x = pre[:, :Co]
        elif pooled and _gap_ok(buf):
            bn = features.norm5
            B, C, H, W = buf.shape
            out = torch.empty((B, C), device=buf.device, dtype=torch.float32)
            check(_lib.lib().mcl_bn_gap_fwd(buf.data_ptr(), C, B, H * W, C, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                            bn.running_mean.data_ptr(), rs[id(bn)].data_ptr(), out.data_ptr(), None,
                                            _stream()), "mcl_bn_gap_fwd")
        else:
            out = affine(buf, features.norm5, False)
            if pooled:
                from . import conv_generic as cg
                out = cg.global_avg_pool(out)               # csrc/pool_generic.hip (fp32 result)
        i += 1
    return out

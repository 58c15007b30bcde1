"""ViT image encoder (/root/reference/model.py:104-116: timm ``vit_base_patch{32,16}_224``,
``num_classes=0, global_pool='avg'``) executed on the hand-written bf16 kernels: every dense contraction -- patch
embedding, QKV / projection / MLP linears, the attention products, all data and weight gradients -- is
``mcl_gemm_bf16`` (csrc/gemm_bf16.hip); LayerNorm, softmax, bias-gradient column sums and patch extraction are the row
kernels of csrc/vit_ops.hip.  Same module tree and parameters as ``backbones.VisionTransformer`` (timm layout: reference
checkpoints load); only the execution differs.  Forward AND hand-scheduled backward of the whole encoder are one
``torch.autograd.Function`` -- the reference obtains the backward from autograd (train.py:38).

Activations are bf16 (B, T, D) with T = patches + 1 (class token first); parameters stay fp32 masters, the kernels read
bf16 copies (FusedAdam's flat shadow when one is attached, densenet_fused._weight).  Attention scores are materialised in
bf16 (heads x T x T per image, rows padded to a multiple of 16): < 1 % of the encoder's FLOPs at T = 197, so one GEMM
kernel serves everything.  Weight / bias / LayerNorm parameter gradients are deterministic (split-K slabs and column
partials merged in fixed order) and go straight into ``.grad`` when the parameter owns a dense fp32 one.
"""
from __future__ import annotations

from typing import List, Optional

import os
import torch

from . import _lib, ops
from ._lib import check
from . import densenet_fused as _dn
from .densenet_fused import _direct_grad_ok, _ws

Tensor = torch.Tensor
BF = torch.bfloat16
A_KM, B_KM, GELU, GELU_BWD, OUT_F32 = 1, 2, 4, 8, 16
GELU_GRAD_OUT, AUX_IS_GRAD = 32, 64           # fc1 stores gelu'(pre-activation); the data gradient multiplies by it directly


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[Tensor], off: int = 0) -> Optional[int]:
    return None if t is None else t.data_ptr() + off * t.element_size()


def gemm(A: Tensor, B: Tensor, C: Tensor, M: int, N: int, K: int, lda: int, ldb: int, ldc: int, *, flags: int = 0,
         a_off: int = 0, b_off: int = 0, c_off: int = 0, batch: int = 1, batch2: int = 1, sA=(0, 0), sB=(0, 0), sC=(0, 0),
         alpha: float = 1.0, bias: Optional[Tensor] = None, resid: Optional[Tensor] = None, ldr: int = 0, sRb: int = 0,
         r_off: int = 0, aux: Optional[Tensor] = None, ldaux: int = 0, pre_out: Optional[Tensor] = None, ldp: int = 0,
         ksplit: int = 1, accumulate: bool = False) -> None:
    """mcl_gemm_bf16 with element offsets / strides (see include/mclstexp_hip.h)."""
    L = _lib.lib()
    ws = None
    if ksplit > 1:
        ws = _ws(L.mcl_gemm_bf16_workspace_floats(M, ldc, ksplit), C.device)
    check(L.mcl_gemm_bf16(_ptr(A, a_off), lda, sA[0], _ptr(B, b_off), ldb, sB[0], _ptr(C, c_off), ldc, sC[0], M, N, K,
                          batch, batch2, sA[1], sB[1], sC[1], alpha, flags, _ptr(bias), _ptr(resid, r_off), ldr, sRb,
                          _ptr(aux), ldaux, _ptr(pre_out), ldp, ksplit, _ptr(ws), int(accumulate), _st()),
          "mcl_gemm_bf16")


def _w16(w: Tensor) -> Tensor:
    """bf16 copy of a 2-D weight: FusedAdam's flat shadow view when one is attached (one cast per step for all weights),
    else a cast."""
    prov = _dn._weight_provider
    if prov is not None:
        v = prov(w, BF)
        if v is not None and v.is_contiguous():
            return v
    if w.is_contiguous():
        v = _dn.cast_dense_bf16(w)
        if v is not None:
            return v
    return w.detach().to(BF).contiguous()


def _dense_f32(p: Tensor) -> Tensor:
    """The parameter's fp32 values as a dense array (a view; parameters are contiguous fp32, also inside FusedAdam's flat buffer)."""
    t = p.detach()
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("vit_fused: expected a contiguous fp32 parameter")
    return t


def _patch_weight_matrix(w: Tensor, dt: torch.dtype) -> Tensor:
    """timm PatchEmbed's Conv2d weight (D, 3, p, p) as the (D, 3 p p) GEMM operand in (c, iy, ix) column order, whatever memory
    format the parameter has (``model.to(memory_format=channels_last)`` permutes 4-D parameters), in ``dt`` (fp32 / bf16):
    the parameter itself / FusedAdam's shadow when that already is the matrix, else one own strided pack launch."""
    D, C, p, _ = w.shape
    wd = w.detach()
    if wd.is_contiguous():
        if dt == torch.float32:
            return wd.reshape(D, C * p * p)
        return _w16(wd.reshape(D, C * p * p))
    out = torch.empty((D, C * p * p), device=w.device, dtype=dt)
    s0, s1, s2, s3 = wd.stride()
    check(_lib.lib().mcl_strided4_f32(wd.data_ptr(), D, C, p, p, s0, s1, s2, s3, out.data_ptr(), C * p * p, p * p, p, 1,
                                      0 if dt == torch.float32 else 1, 0, _st()), "mcl_strided4_f32")
    return out


def _patch_weight_grad(dy: Tensor, patches: Tensor, w: Tensor, rows: int):
    """d pe.weight from dy (rows, D) and the token-matrix patches (rows, K0).  Straight into ``w.grad`` when that is a dense
    (D, K0) array; for a channels-last parameter the contiguous result is ADDED into the .grad's own strides by one own launch
    (no AccumulateGrad add / layout clone); otherwise returned to autograd."""
    D, C, p, _ = w.shape
    if w.is_contiguous():
        gw = linear_wgrad(dy, patches, w, rows)
        return None if gw is None else gw.view_as(w)
    K0 = C * p * p
    gw = torch.empty((D, K0), device=dy.device, dtype=torch.float32)
    ks = _ksplit(D, K0)
    gemm(dy, patches, gw, D, K0, rows, dy.shape[-1], K0, K0, flags=A_KM | B_KM | OUT_F32, ksplit=ks, accumulate=False)
    if _direct_grad_ok(w):
        g = w.grad
        s0, s1, s2, s3 = g.stride()
        check(_lib.lib().mcl_strided4_f32(gw.data_ptr(), D, C, p, p, K0, p * p, p, 1, g.data_ptr(), s0, s1, s2, s3, 0, 1, _st()),
              "mcl_strided4_f32")
        return None
    return gw.view(D, C, p, p)


GELU_HANDOFF = True         # 0: fc1 stores the pre-activation, gelu' evaluated in the backward (A/B)
FUSED_ATTN = True            # 0: batched GEMMs + softmax launches (A/B, T > 224)
JOIN_EVERY = 4              # encoder blocks between joins of the side stream
KSPLIT_TARGET = 128     # workgroups a split-K weight gradient aims for (side lane: 60.5 ms/step at 256, 59.7 at 128)


def _ksplit(m_out: int, n_out: int) -> int:
    tiles = ((m_out + 255) // 256) * ((n_out + 255) // 256)
    return max(2, min(64, (KSPLIT_TARGET + tiles - 1) // tiles))


def _grad_target(p: Tensor):
    """(tensor to write, accumulate?, value to return to autograd)."""
    if _direct_grad_ok(p) and p.grad.is_contiguous():
        return p.grad, True, None
    g = torch.empty_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
    return g, False, g


# Weight and bias gradients only feed the optimizer: they run on the side stream (the step's second lane, shared with the
# spot branch and the DenseNet weight gradients) while the data-gradient chain continues, whenever they accumulate
# straight into the parameters' .grad (nothing is handed back to autograd from the other stream).
SIDE_WGRAD = True


def _param_grads(dy: Tensor, x: Tensor, lin, rows: int, grads: dict) -> None:
    """grads[lin.weight], grads[lin.bias] for y = x W^T + b given dy; on the side stream when both are written in place."""
    from . import densenet_fused as dn
    direct = _grad_target(lin.weight)[1] and (lin.bias is None or _grad_target(lin.bias)[1])
    if SIDE_WGRAD and dn.USE_SIDE_STREAM and direct:
        main = torch.cuda.current_stream()
        side = dn._side_stream(dy.device)
        side.wait_stream(main)                # dy is final on the main stream
        with torch.cuda.stream(side):
            grads[lin.weight] = linear_wgrad(dy, x, lin.weight, rows)
            if lin.bias is not None:
                grads[lin.bias] = bias_grad(dy, lin.bias, rows)
        dn._side_park(dy.device, dy, x)
        return
    grads[lin.weight] = linear_wgrad(dy, x, lin.weight, rows)
    if lin.bias is not None:
        grads[lin.bias] = bias_grad(dy, lin.bias, rows)


def linear_wgrad(dy: Tensor, x: Tensor, w: Tensor, rows: int):
    """dW[out][in] (+)= dy^T x over ``rows`` tokens (both operands reduction-major, split-K, fixed-order merge)."""
    n_out, n_in = w.shape[0], w[0].numel()
    tgt, acc, ret = _grad_target(w)
    gemm(dy, x, tgt, n_out, n_in, rows, dy.shape[-1], x.shape[-1], n_in, flags=A_KM | B_KM | OUT_F32,
         ksplit=_ksplit(n_out, n_in), accumulate=acc)
    return ret


def bias_grad(dy: Tensor, b: Tensor, rows: int):
    D = b.numel()
    tgt, acc, ret = _grad_target(b)
    L = _lib.lib()
    ws = _ws(L.mcl_colred_workspace_floats(rows, D), dy.device)
    check(L.mcl_colsum_bf16(dy.data_ptr(), dy.shape[-1], rows, D, ws.data_ptr(), tgt.data_ptr(), int(acc), _st()),
          "mcl_colsum_bf16")
    return ret


def ln_fwd(x: Tensor, ln: torch.nn.LayerNorm, rows: int):
    D = x.shape[-1]
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    check(_lib.lib().mcl_ln_bf16_fwd(x.data_ptr(), D, ln.weight.data_ptr(), ln.bias.data_ptr(), y.data_ptr(), D,
                                     mean.data_ptr(), rstd.data_ptr(), rows, D, float(ln.eps), _st()), "mcl_ln_bf16_fwd")
    return y, mean, rstd


def ln_bwd(dy: Tensor, x: Tensor, ln: torch.nn.LayerNorm, mean: Tensor, rstd: Tensor, dx_add: Optional[Tensor], rows: int):
    """(dx [+ dx_add], dgamma-return, dbeta-return)."""
    D = x.shape[-1]
    dx = torch.empty_like(x)
    tg, acc_g, ret_g = _grad_target(ln.weight)
    tb, acc_b, ret_b = _grad_target(ln.bias)
    assert acc_g == acc_b
    L = _lib.lib()
    ws = _ws(L.mcl_colred_workspace_floats(rows, D), x.device)
    check(L.mcl_ln_bf16_bwd(dy.data_ptr(), D, x.data_ptr(), D, ln.weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                            _ptr(dx_add), D if dx_add is not None else 0, dx.data_ptr(), D, ws.data_ptr(), tg.data_ptr(),
                            tb.data_ptr(), int(acc_g), rows, D, _st()), "mcl_ln_bf16_bwd")
    return dx, ret_g, ret_b


class ViTFn(torch.autograd.Function):
    """features (B, D) fp32 = VisionTransformer(image) up to and including the mean pool over the patch tokens (the
    final ``fc_norm`` runs on ops.LayerNormFn, fp32).  ``vit`` (the module) rides along for shapes and parameter
    objects; its parameters are passed explicitly so that autograd routes their gradients."""

    @staticmethod
    def forward(ctx, image, vit, *params):
        dev = image.device
        pe = vit.patch_embed.proj
        p = pe.kernel_size[0]
        B, _, H, W = image.shape
        nph, npw = H // p, W // p
        npatch, T = nph * npw, nph * npw + 1
        D = pe.out_channels
        K0 = 3 * p * p
        heads = vit.blocks[0].attn.num_heads
        dh = D // heads
        M = B * T
        Tp = (T + 15) // 16 * 16
        L = _lib.lib()
        img = image if image.dtype == torch.float32 else image.float()
        # patches with a zero row at the class-token position: (B, T, K0); the weight gradient of the patch embedding
        # is then ONE reduction over all B*T rows.  Token assembly on own kernels (csrc/glue.hip): no ATen launch.
        patches = torch.empty((B, T, K0), device=dev, dtype=BF)
        check(L.mcl_vit_patchify_tokens(img.data_ptr(), img.stride(0), img.stride(1), img.stride(2), img.stride(3), B, H, W, p,
                                        patches.data_ptr(), 1, 0, _st()), "mcl_vit_patchify_tokens")
        pos, cls = _dense_f32(vit.pos_embed), _dense_f32(vit.cls_token)
        pos16 = torch.empty((T, D), device=dev, dtype=BF)
        check(L.mcl_cast_f32_to_bf16(pos.data_ptr(), D, pos16.data_ptr(), D, T, D, _st()), "mcl_cast_f32_to_bf16")
        x = torch.empty((B, T, D), device=dev, dtype=BF)
        wpe = _patch_weight_matrix(pe.weight, BF)                            # (D, K0), (c, iy, ix) order whatever the memory format
        gemm(patches, wpe, x, npatch, D, K0, K0, K0, D, a_off=K0, c_off=D, batch=B, sA=(T * K0, 0), sC=(T * D, 0),
             bias=pe.bias, resid=pos16, ldr=D, sRb=0, r_off=D)
        check(L.mcl_vit_cls_row(cls.data_ptr(), pos.data_ptr(), x.data_ptr(), B, T, D, 1, _st()), "mcl_vit_cls_row")
        saved = []
        scale = dh ** -0.5
        fused_attn = FUSED_ATTN and dh == 64 and T <= 224
        for blk in vit.blocks:
            a, m = blk.attn, blk.mlp
            u1, mean1, rstd1 = ln_fwd(x, blk.norm1, M)
            qkv = torch.empty((B, T, 3 * D), device=dev, dtype=BF)
            gemm(u1, _w16(a.qkv.weight), qkv, M, 3 * D, D, D, D, 3 * D, bias=a.qkv.bias)
            o = torch.empty((B, T, D), device=dev, dtype=BF)
            if fused_attn:
                # softmax(q k^T scale) v per image and head in ONE launch, no (B heads, T, T) tensor (csrc/vit_attention.hip);
                # P below is the row log-sum-exp, all the backward needs
                P = torch.empty((B * heads, T), device=dev, dtype=torch.float32)
                check(L.mcl_vit_attn_fwd(qkv.data_ptr(), o.data_ptr(), P.data_ptr(), B, T, heads, scale, _st()), "mcl_vit_attn_fwd")
            else:
                P = torch.empty((B * heads, T, Tp), device=dev, dtype=BF)
                gemm(qkv, qkv, P, T, T, dh, 3 * D, 3 * D, Tp, b_off=D, batch=B * heads, batch2=heads,
                     sA=(T * 3 * D, dh), sB=(T * 3 * D, dh), sC=(heads * T * Tp, T * Tp), alpha=scale)
                check(L.mcl_softmax_bf16_fwd(P.data_ptr(), Tp, B * heads * T, T, _st()), "mcl_softmax_bf16_fwd")
                gemm(P, qkv, o, T, dh, T, Tp, 3 * D, D, flags=B_KM, b_off=2 * D, batch=B * heads, batch2=heads,
                     sA=(heads * T * Tp, T * Tp), sB=(T * 3 * D, dh), sC=(T * D, dh))
            x1 = torch.empty_like(x)
            gemm(o, _w16(a.proj.weight), x1, M, D, D, D, D, D, bias=a.proj.bias, resid=x, ldr=D, sRb=0)
            u2, mean2, rstd2 = ln_fwd(x1, blk.norm2, M)
            Dh = m.fc1.out_features
            h1 = torch.empty((B, T, Dh), device=dev, dtype=BF)
            pre = torch.empty((B, T, Dh), device=dev, dtype=BF)
            gemm(u2, _w16(m.fc1.weight), h1, M, Dh, D, D, D, Dh, flags=GELU | (GELU_GRAD_OUT if GELU_HANDOFF else 0), bias=m.fc1.bias, pre_out=pre, ldp=Dh)
            x2 = torch.empty_like(x)
            gemm(h1, _w16(m.fc2.weight), x2, M, D, Dh, Dh, Dh, D, bias=m.fc2.bias, resid=x1, ldr=D, sRb=0)
            saved += [x, mean1, rstd1, u1, qkv, P, o, x1, mean2, rstd2, u2, pre, h1]
            x = x2
        feat = torch.empty((B, D), device=dev, dtype=torch.float32)      # global_pool='avg' over the patch tokens
        check(L.mcl_vit_token_mean_fwd(x.data_ptr(), feat.data_ptr(), B, T, D, 1, _st()), "mcl_vit_token_mean_fwd")
        ctx.save_for_backward(patches, *saved)
        ctx.vit = vit
        ctx.dims = (B, T, D, K0, heads, dh, Tp, npatch)
        ctx.fused_attn = fused_attn
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        vit = ctx.vit
        B, T, D, K0, heads, dh, Tp, npatch = ctx.dims
        M = B * T
        t = ctx.saved_tensors
        patches, saved = t[0], t[1:]
        dev = dfeat.device
        L = _lib.lib()
        from . import densenet_fused as _dn
        scale = dh ** -0.5
        dx = torch.empty((B, T, D), device=dev, dtype=BF)
        dfeat = dfeat if (dfeat.dtype == torch.float32 and dfeat.is_contiguous()) else dfeat.float().contiguous()
        check(L.mcl_vit_token_mean_bwd(dfeat.data_ptr(), dx.data_ptr(), B, T, D, 1, _st()), "mcl_vit_token_mean_bwd")
        grads = {}
        nblk = len(vit.blocks)
        for li in range(nblk - 1, -1, -1):
            blk = vit.blocks[li]
            a, m = blk.attn, blk.mlp
            x, mean1, rstd1, u1, qkv, P, o, x1, mean2, rstd2, u2, pre, h1 = saved[13 * li: 13 * li + 13]
            Dh = m.fc1.out_features
            # MLP
            _param_grads(dx, h1, m.fc2, M, grads)
            dpre = torch.empty((B, T, Dh), device=dev, dtype=BF)
            gemm(dx, _w16(m.fc2.weight), dpre, M, Dh, D, D, Dh, Dh, flags=B_KM | (AUX_IS_GRAD if GELU_HANDOFF else GELU_BWD), aux=pre, ldaux=Dh)   # pre = gelu' (hand-off)
            _param_grads(dpre, u2, m.fc1, M, grads)
            du2 = torch.empty((B, T, D), device=dev, dtype=BF)
            gemm(dpre, _w16(m.fc1.weight), du2, M, D, Dh, Dh, D, D, flags=B_KM)
            dx1, grads[blk.norm2.weight], grads[blk.norm2.bias] = ln_bwd(du2, x1, blk.norm2, mean2, rstd2, dx, M)
            # attention output projection
            _param_grads(dx1, o, a.proj, M, grads)
            do = torch.empty((B, T, D), device=dev, dtype=BF)
            gemm(dx1, _w16(a.proj.weight), do, M, D, D, D, D, D, flags=B_KM)
            # attention core, per (image, head)
            dqkv = torch.empty((B, T, 3 * D), device=dev, dtype=BF)
            nb = B * heads
            if ctx.fused_attn:
                # P = the forward's row log-sum-exp: dq, dk, dv in two launches, probabilities recomputed on chip
                dsum = torch.empty((nb, T), device=dev, dtype=torch.float32)
                check(L.mcl_vit_attn_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), P.data_ptr(), dsum.data_ptr(),
                                         dqkv.data_ptr(), B, T, heads, scale, _st()), "mcl_vit_attn_bwd")
            else:
                sP = (heads * T * Tp, T * Tp)
                sQ = (T * 3 * D, dh)
                sO = (T * D, dh)
                gemm(P, do, dqkv, T, dh, T, Tp, D, 3 * D, flags=A_KM | B_KM, c_off=2 * D, batch=nb, batch2=heads,
                     sA=sP, sB=sO, sC=sQ)                                                   # dV = P^T dO
                dP = torch.empty((B * heads, T, Tp), device=dev, dtype=BF)
                gemm(do, qkv, dP, T, T, dh, D, 3 * D, Tp, b_off=2 * D, batch=nb, batch2=heads, sA=sO, sB=sQ, sC=sP)   # dO V^T
                check(L.mcl_softmax_bf16_bwd(P.data_ptr(), dP.data_ptr(), Tp, B * heads * T, T, scale, _st()),
                      "mcl_softmax_bf16_bwd")
                gemm(dP, qkv, dqkv, T, dh, T, Tp, 3 * D, 3 * D, flags=B_KM, b_off=D, batch=nb, batch2=heads,
                     sA=sP, sB=sQ, sC=sQ)                                                   # dQ = dS K
                gemm(dP, qkv, dqkv, T, dh, T, Tp, 3 * D, 3 * D, flags=A_KM | B_KM, c_off=D, batch=nb, batch2=heads,
                     sA=sP, sB=sQ, sC=sQ)                                                   # dK = dS^T Q
            _param_grads(dqkv, u1, a.qkv, M, grads)
            du1 = torch.empty((B, T, D), device=dev, dtype=BF)
            gemm(dqkv, _w16(a.qkv.weight), du1, M, D, 3 * D, 3 * D, D, D, flags=B_KM)
            dx, grads[blk.norm1.weight], grads[blk.norm1.bias] = ln_bwd(du1, x, blk.norm1, mean1, rstd1, dx1, M)
            if li % JOIN_EVERY == 0:
                _dn._side_join(dev)      # bounds what stays parked for the side stream (0.7 GB of operands per block)
        # embeddings: position table and class token (fp32 sums over the batch), patch projection
        tp, accp, retp = _grad_target(vit.pos_embed)
        tc, accc, retc = _grad_target(vit.cls_token)
        check(L.mcl_vit_pos_grad(dx.data_ptr(), tp.data_ptr(), tc.data_ptr(), B, T, D, 1, (1 if accp else 0) | (2 if accc else 0),
                                 _st()), "mcl_vit_pos_grad")
        grads[vit.pos_embed], grads[vit.cls_token] = retp, retc
        pe = vit.patch_embed.proj
        # class-token rows carry no patch: zeroed in place (dx is dead after this) for the bias gradient; their patch rows are zero
        check(L.mcl_vit_zero_cls_rows(dx.data_ptr(), B, T, D, 1, _st()), "mcl_vit_zero_cls_rows")
        grads[pe.weight] = _patch_weight_grad(dx, patches, pe.weight, M)
        grads[pe.bias] = bias_grad(dx, pe.bias, M)
        _dn._side_join(dev)
        return (None, None, *[grads.get(prm) for prm in _param_list(vit)])


class _EmbedF32Fn(torch.autograd.Function):
    """timm PatchEmbed + class token + position embedding in fp32: x (B*T, D) = [cls ; patches W^T + b] + pos, on own kernels
    (patch unfold, exact-fp32 MFMA product, assembly; the backward's token extraction, weight / bias / position / class-token
    gradients -- written straight into dense .grad buffers where the parameters own them)."""

    @staticmethod
    def forward(ctx, img, w, b, cls, pos, p):
        L = _lib.lib()
        B, Cin, H, W = img.shape
        nph, npw = H // p, W // p
        npatch, T = nph * npw, nph * npw + 1
        D, K0 = w.shape[0], Cin * p * p
        dev = img.device
        patches = torch.empty((B * npatch, K0), device=dev, dtype=torch.float32)
        check(L.mcl_vit_patchify_tokens(img.data_ptr(), img.stride(0), img.stride(1), img.stride(2), img.stride(3), B, H, W, p,
                                        patches.data_ptr(), 0, 1, _st()), "mcl_vit_patchify_tokens")
        wm = _patch_weight_matrix(w, torch.float32)
        with ops.forced_compute(_lib.COMPUTE_F32):
            tok, _ = ops.linear_fwd(patches, wm, b.detach() if b is not None else None)
        x = torch.empty((B * T, D), device=dev, dtype=torch.float32)
        check(L.mcl_vit_assemble_f32(tok.data_ptr(), _dense_f32(cls).data_ptr(), _dense_f32(pos).data_ptr(), x.data_ptr(), B, T, D,
                                     _st()), "mcl_vit_assemble_f32")
        ctx.save_for_backward(patches)
        ctx.params = (w, b, cls, pos)
        ctx.dims = (B, T, D, Cin, p)
        return x

    @staticmethod
    def backward(ctx, dx):
        (patches,) = ctx.saved_tensors
        w, b, cls, pos = ctx.params
        B, T, D, Cin, p = ctx.dims
        L = _lib.lib()
        dx = ops._rowmajor(dx, "dx")
        if dx.stride(0) != D:
            raise RuntimeError("vit_fused: the token gradient must be a dense (B*T, D) matrix")
        dtok = torch.empty((B * (T - 1), D), device=dx.device, dtype=torch.float32)
        check(L.mcl_vit_tokens_extract(dx.data_ptr(), dtok.data_ptr(), B, T, D, 0, _st()), "mcl_vit_tokens_extract")
        tp, accp, retp = _grad_target(pos)
        tc, accc, retc = _grad_target(cls)
        check(L.mcl_vit_pos_grad(dx.data_ptr(), tp.data_ptr(), tc.data_ptr(), B, T, D, 0, (1 if accp else 0) | (2 if accc else 0),
                                 _st()), "mcl_vit_pos_grad")
        with ops.forced_compute(_lib.COMPUTE_F32):
            K0 = Cin * p * p
            if w.is_contiguous() and _direct_grad_ok(w) and w.grad.is_contiguous():
                ops.gemm_raw(D, K0, dtok.shape[0], 1, dtok, 1, D, 0, patches, K0, 1, 0, w.grad, K0, 0, flags=ops.EPI_ACCUM)
                dw = None
            else:
                gw = torch.empty((D, K0), device=dx.device, dtype=torch.float32)
                ops.gemm_raw(D, K0, dtok.shape[0], 1, dtok, 1, D, 0, patches, K0, 1, 0, gw, K0, 0)
                if _direct_grad_ok(w):
                    g = w.grad
                    s0, s1, s2, s3 = g.stride()
                    check(L.mcl_strided4_f32(gw.data_ptr(), D, Cin, p, p, K0, p * p, p, 1, g.data_ptr(), s0, s1, s2, s3, 0, 1, _st()),
                          "mcl_strided4_f32")
                    dw = None
                else:
                    dw = gw.view(D, Cin, p, p)
            db = ops.colsum(dtok, b) if b is not None else None
        return None, dw, db, retc, retp, None


class _TokenMeanF32Fn(torch.autograd.Function):
    """timm's global_pool='avg': mean over the patch tokens (class token excluded) of x (B*T, D) fp32 -> (B, D)."""

    @staticmethod
    def forward(ctx, x, B, T):
        x = ops._rowmajor(x, "x")
        D = x.shape[1]
        if x.stride(0) != D:
            raise RuntimeError("vit_fused: the token matrix must be dense")
        feat = torch.empty((B, D), device=x.device, dtype=torch.float32)
        check(_lib.lib().mcl_vit_token_mean_fwd(x.data_ptr(), feat.data_ptr(), B, T, D, 0, _st()), "mcl_vit_token_mean_fwd")
        ctx.dims = (B, T, D)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        B, T, D = ctx.dims
        dfeat = dfeat if (dfeat.dtype == torch.float32 and dfeat.is_contiguous()) else dfeat.float().contiguous()
        dx = torch.empty((B * T, D), device=dfeat.device, dtype=torch.float32)
        check(_lib.lib().mcl_vit_token_mean_bwd(dfeat.data_ptr(), dx.data_ptr(), B, T, D, 0, _st()), "mcl_vit_token_mean_bwd")
        return dx, None, None


def vit_features_fp32(vit, image: Tensor) -> Tensor:
    """``VisionTransformer.forward`` in fp32 ("reference numerics": /root/reference/model.py:104-116 computes in fp32) on this
    library's fp32 kernels: every contraction is ``mcl_gemm`` in exact-fp32 MFMA mode, LayerNorm / GELU / bias gradients the
    spot branch's kernels, the attention core csrc/attention.hip with one sequence per image (no (B, heads, T, T) tensor).
    Token assembly (unfold of the patches, class token, position embedding) and the final token mean are own kernels as well
    (csrc/glue.hip; round 6).  (B, D) fp32."""
    if not image.is_cuda:
        raise RuntimeError("vit_features_fp32: input is on the CPU; the ViT kernels are GPU-only")
    F32 = _lib.COMPUTE_F32
    pe = vit.patch_embed.proj
    p = pe.kernel_size[0]
    B, Cin, H, W = image.shape
    nph, npw = H // p, W // p
    npatch, T = nph * npw, nph * npw + 1
    D = pe.out_channels
    heads = vit.blocks[0].attn.num_heads
    dh = D // heads
    if dh != 64:
        raise RuntimeError(f"vit_features_fp32: head dimension {dh} (the fp32 attention kernel is built for 64)")
    img = image if image.dtype == torch.float32 else image.float()
    x = _EmbedF32Fn.apply(img, pe.weight, pe.bias, vit.cls_token, vit.pos_embed, p)          # (B*T, D), own kernels only
    for blk in vit.blocks:
        a, m = blk.attn, blk.mlp
        x = ops.AttnBlockFn.apply(x, blk.norm1.weight, blk.norm1.bias, a.qkv.weight, a.proj.weight, a.proj.bias,
                                  blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                  heads, dh, a.qkv.bias, B, float(blk.norm1.eps), F32)
    feat = _TokenMeanF32Fn.apply(x, B, T)                          # global_pool='avg' over the patch tokens
    return ops.LayerNormFn.apply(feat, vit.fc_norm.weight, vit.fc_norm.bias, vit.fc_norm.eps)


def _param_list(vit) -> List[Tensor]:
    """The encoder's trainable parameters in module order, without fc_norm (which runs outside ViTFn)."""
    skip = {id(vit.fc_norm.weight), id(vit.fc_norm.bias)}
    return [p for p in vit.parameters() if p.requires_grad and id(p) not in skip]


def vit_features_fused(vit, image: Tensor) -> Tensor:
    """``VisionTransformer.forward`` (timm: global_pool='avg', fc_norm) on the bf16 kernels; (B, D) fp32."""
    if not image.is_cuda:
        raise RuntimeError("vit_features_fused: input is on the CPU; the fused ViT path is GPU-only")
    feat = ViTFn.apply(image, vit, *_param_list(vit))
    return ops.LayerNormFn.apply(feat, vit.fc_norm.weight, vit.fc_norm.bias, vit.fc_norm.eps)

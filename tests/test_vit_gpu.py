"""ViT image encoder on the hand-written bf16 kernels (csrc/gemm_bf16.hip, csrc/vit_ops.hip, mclstexp_amd/vit_fused.py)
against fp64 torch on the same bf16 data.  pytest -m gpu."""
import copy

import numpy as np
import pytest
import torch

from helpers import assert_close, assert_close_scaled

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _r(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(*shape, generator=g) - 0.5) * 2 * scale)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (197, 197, 64), (300, 200, 136), (1000, 768, 768), (64, 72, 200)])
@pytest.mark.parametrize("akm,bkm", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_bf16_layouts(M, N, K, akm, bkm):
    """All four operand storage combinations, ragged M / N / K, leading dimensions wider than the matrix."""
    from mclstexp_amd import vit_fused as vf
    lda = ((M if akm else K) + 7) // 8 * 8 + 8
    ldb = ((N if bkm else K) + 7) // 8 * 8 + 16
    ldc = (N + 7) // 8 * 8 + 8
    A = _r(K if akm else M, lda, seed=1).to(BF).to(DEV)
    B = _r(K if bkm else N, ldb, seed=2).to(BF).to(DEV)
    C = torch.full((M, ldc), 7.0, device=DEV, dtype=BF)
    vf.gemm(A, B, C, M, N, K, lda, ldb, ldc, flags=akm * vf.A_KM | bkm * vf.B_KM, alpha=0.5)
    a = (A[:K, :M].t() if akm else A[:M, :K]).double()
    b = (B[:K, :N] if bkm else B[:N, :K].t()).double()
    ref = 0.5 * (a @ b)
    assert_close(C[:, :N].float().cpu(), ref.cpu(), 2e-3, 2 ** -8, what=f"gemm akm={akm} bkm={bkm}")
    if ldc - (N + 7) // 8 * 8 > 0:
        assert (C[:, (N + 7) // 8 * 8:] == 7.0).all()               # columns beyond the padded row are untouched


@pytest.mark.parametrize("M,N,K,akm,bkm,mode", [
    (1024, 768, 768, 0, 0, "bias"), (1024, 2304, 64, 0, 0, "plain"), (768, 1024, 128, 0, 1, "gelu_bwd"),
    (1024, 768, 192, 0, 1, "resid"), (2304, 1024, 256, 1, 0, "bias_gelu_pre"), (768, 3072, 8192, 1, 1, "f32_splitk"),
    (1024, 1024, 4096, 1, 1, "f32"),
    # several tiles per persistent workgroup (576 / 1152 tiles on 256 / 512 workgroups), the ViT block's own epilogue forms
    (16384, 2304, 768, 0, 0, "bias_gelu2"), (16384, 2304, 256, 0, 1, "gelu_bwd2"), (16384, 2304, 384, 0, 0, "resid"),
    (16384, 2304, 128, 1, 0, "bias"), (2304, 768, 16384, 1, 1, "f32_splitk")])
def test_gemm_bf16_staggered_kernel_bit_identical(M, N, K, akm, bkm, mode, monkeypatch):
    """gemm_bf16_stag_kernel (problems of interior 256 x 256 tiles: two wave groups one barrier apart over a four-slot half-K
    ring) against the lockstep kernel (MCL_GEMM_STAG=0): same fragments in the same k order and the same epilogue, so equal
    bit for bit -- all four operand layouts, 1 / 2 / 3 / many half-tiles (shorter than, equal to and longer than the ring),
    every epilogue form, split-K slabs; and against fp64."""
    from mclstexp_amd import vit_fused as vf
    A = _r(K if akm else M, M if akm else K, seed=21).to(BF).to(DEV)
    B = _r(K if bkm else N, N if bkm else K, seed=22, scale=0.3).to(BF).to(DEV)
    bias = _r(N, seed=23).to(DEV)
    res = _r(M, N, seed=24).to(BF).to(DEV)
    aux = _r(M, N, seed=25, scale=2.0).to(BF).to(DEV)
    flags = akm * vf.A_KM | bkm * vf.B_KM
    outs = []
    # staggered, lockstep (the reference of the bit comparison), then the round-6 persistent pipelined kernel (it declines K ranges
    # under 4 half-tiles: those launches fall through to the staggered one)
    for stag, pipe in (("1", "0"), ("0", "0"), ("1", "1")):
        monkeypatch.setenv("MCL_GEMM_STAG", stag)
        monkeypatch.setenv("MCL_GEMM_PIPE", pipe)
        f32 = mode.startswith("f32")
        C = torch.full((M, N), 3.0, device=DEV, dtype=torch.float32 if f32 else BF)
        pre = torch.zeros((M, N), device=DEV, dtype=BF)
        kw = dict(flags=flags)
        if mode == "plain":
            kw.update(alpha=0.5)
        elif mode == "bias":
            kw.update(bias=bias)
        elif mode == "bias_gelu_pre":
            kw.update(flags=flags | vf.GELU, bias=bias, pre_out=pre, ldp=N)
        elif mode == "bias_gelu2":
            kw.update(flags=flags | vf.GELU | vf.GELU_GRAD_OUT, bias=bias, pre_out=pre, ldp=N)
        elif mode == "gelu_bwd2":
            kw.update(flags=flags | vf.AUX_IS_GRAD, aux=aux, ldaux=N)
        elif mode == "resid":
            kw.update(bias=bias, resid=res, ldr=N)
        elif mode == "gelu_bwd":
            kw.update(flags=flags | vf.GELU_BWD, aux=aux, ldaux=N)
        elif mode == "f32":
            kw.update(flags=flags | vf.OUT_F32)
        else:
            kw.update(flags=flags | vf.OUT_F32, ksplit=8, accumulate=True)
        vf.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N, **kw)
        torch.cuda.synchronize()
        outs.append((C, pre))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    for k in (2,):
        # same fragments and k order, MFMA operands swapped (transposed result layout): the products are equal bit for bit; the
        # GELU epilogues are compiled in a different expression context (fused multiply-adds may pair differently): one bf16 ulp
        for w in (0, 1):
            if "gelu" in mode:
                d = (outs[k][w].float() - outs[1][w].float()).abs()
                assert bool((d <= 2.0 ** -7 * outs[1][w].float().abs() + 1e-30).all()), (k, w, float(d.max()))
                assert float((d > 0).float().mean()) < 0.02
            else:
                assert torch.equal(outs[k][w], outs[1][w]), (k, w)
    a = (A.t() if akm else A).double()
    b = (B if bkm else B.t()).double()
    ref = a @ b
    if mode == "plain":
        ref = 0.5 * ref
    elif mode == "bias":
        ref = ref + bias.double()
    elif mode == "bias_gelu_pre":
        assert_close(outs[0][1].float().cpu(), (ref + bias.double()).cpu(), 3e-3, 2 ** -7, what="staggered pre-activation")
        ref = torch.nn.functional.gelu(ref + bias.double())
    elif mode == "bias_gelu2":
        xa = (ref + bias.double()).requires_grad_(True)
        y = torch.nn.functional.gelu(xa)
        y.sum().backward()
        assert_close(outs[0][1].float().cpu(), xa.grad.cpu(), 3e-3, 2 ** -7, what="stored gelu'")
        ref = y.detach()
    elif mode == "gelu_bwd2":
        ref = ref * aux.double()
    elif mode == "resid":
        ref = ref + bias.double() + res.double()
    elif mode == "gelu_bwd":
        xa = aux.double().requires_grad_(True)
        torch.nn.functional.gelu(xa).sum().backward()
        ref = ref * xa.grad
    elif mode == "f32_splitk":
        ref = ref + 3.0
    tol = (2e-4, 2e-3) if mode.startswith("f32") else (4e-3, 2 ** -6)
    assert_close(outs[0][0].float().cpu(), ref.cpu(), tol[0], tol[1], what=f"staggered gemm {mode}")


@pytest.mark.parametrize("B,T,heads", [(2, 197, 12), (3, 50, 2), (2, 33, 1), (1, 224, 2), (2, 7, 3)])
def test_vit_fused_attention_vs_fp64(B, T, heads):
    """csrc/vit_attention.hip (softmax(q k^T scale) v per image and head, forward + dq / dk / dv, no probability tensor in HBM)
    against fp64 autograd on the same bf16 qkv / dout values: o and the row log-sum-exp, and the three gradients -- ragged
    token counts (197, 50, 33, 7: partial 32-blocks), the 224-token limit, several heads.  bf16 outputs: tolerances are the
    output rounding (2^-8 relative) plus the bf16 rounding of P / dS inside the second product."""
    from mclstexp_amd import _lib
    from mclstexp_amd._lib import check
    L = _lib.lib()
    D = heads * 64
    scale = 64 ** -0.5
    qkv = _r(B, T, 3 * D, seed=31, scale=2.0).to(BF).to(DEV)
    dout = _r(B, T, D, seed=32).to(BF).to(DEV)
    o = torch.zeros((B, T, D), device=DEV, dtype=BF)
    lse = torch.zeros((B * heads, T), device=DEV, dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    check(L.mcl_vit_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, T, heads, scale, st), "fwd")
    dqkv = torch.full((B, T, 3 * D), 7.0, device=DEV, dtype=BF)
    dsum = torch.zeros((B * heads, T), device=DEV, dtype=torch.float32)
    check(L.mcl_vit_attn_bwd(qkv.data_ptr(), o.data_ptr(), dout.data_ptr(), lse.data_ptr(), dsum.data_ptr(), dqkv.data_ptr(),
                             B, T, heads, scale, st), "bwd")
    torch.cuda.synchronize()
    x = qkv.double().cpu().requires_grad_(True)
    q, k, v = [x[..., i * D:(i + 1) * D].reshape(B, T, heads, 64).permute(0, 2, 1, 3) for i in range(3)]
    s = (q @ k.transpose(-1, -2)) * scale
    ref_lse = torch.logsumexp(s, dim=-1)
    ref_o = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, T, D)
    ref_o.backward(dout.double().cpu())
    om = float(ref_o.detach().abs().max())
    assert_close(o.float().cpu(), ref_o.detach(), 6e-3 * om, 2 ** -7, what="o")
    assert_close(lse.cpu(), ref_lse.detach().reshape(B * heads, T), 2e-4, 1e-5, what="lse")
    g = x.grad
    for i, name in enumerate(("dq", "dk", "dv")):
        ref = g[..., i * D:(i + 1) * D]
        mx = float(ref.abs().max())
        assert_close(dqkv[..., i * D:(i + 1) * D].float().cpu(), ref, 1.5e-2 * mx, 2 ** -6, what=name)
    assert torch.isfinite(dqkv.float()).all()


def test_gemm_bf16_gelu_derivative_handoff():
    """fc1 -> fc2 hand-off of the ViT MLP: the forward epilogue stores gelu'(pre-activation) beside gelu(pre-activation) (one
    exp / rcp serves both), the data-gradient epilogue multiplies by that stored derivative (no transcendental per element)."""
    from mclstexp_amd import vit_fused as vf
    M, N, K = 1024, 768, 256
    A = _r(M, K, seed=41).to(BF).to(DEV)
    W = _r(N, K, seed=42, scale=0.3).to(BF).to(DEV)
    bias = _r(N, seed=43).to(DEV)
    h = torch.empty((M, N), device=DEV, dtype=BF)
    gp = torch.empty((M, N), device=DEV, dtype=BF)
    vf.gemm(A, W, h, M, N, K, K, K, N, flags=vf.GELU | vf.GELU_GRAD_OUT, bias=bias, pre_out=gp, ldp=N)
    pre = (A.double() @ W.double().t() + bias.double()).cpu().requires_grad_(True)
    ref_h = torch.nn.functional.gelu(pre)
    ref_h.sum().backward()
    assert_close(h.float().cpu(), ref_h.detach(), 4e-3, 2 ** -7, what="gelu")
    assert_close(gp.float().cpu(), pre.grad, 4e-3, 2 ** -7, what="stored gelu'")
    dy = _r(M, K, seed=44).to(BF).to(DEV)
    W2 = _r(K, N, seed=45, scale=0.3).to(BF).to(DEV)               # dy (M, K) @ W2 (K, N) as the reduction-major operand
    dpre = torch.empty((M, N), device=DEV, dtype=BF)
    vf.gemm(dy, W2, dpre, M, N, K, K, N, N, flags=vf.B_KM | vf.AUX_IS_GRAD, aux=gp, ldaux=N)
    ref = (dy.double() @ W2.double()) * gp.double()
    assert_close(dpre.float().cpu(), ref.cpu(), 4e-3, 2 ** -7, what="dgrad * stored gelu'")


def test_gemm_bf16_epilogues_and_batch():
    """bias + GELU (+ stored pre-activation), gelu' multiply, residual add, two-level batch with strides, fp32 output,
    split-K with accumulation (deterministic)."""
    from mclstexp_amd import vit_fused as vf
    M, N, K = 300, 264, 200
    A = _r(M, K, seed=3).to(BF).to(DEV)
    W = _r(N, K, seed=4, scale=0.2).to(BF).to(DEV)
    bias = _r(N, seed=5).to(DEV)
    res = _r(M, N, seed=6).to(BF).to(DEV)
    pre = torch.empty((M, N), device=DEV, dtype=BF)
    h = torch.empty((M, N), device=DEV, dtype=BF)
    vf.gemm(A, W, h, M, N, K, K, K, N, flags=vf.GELU, bias=bias, pre_out=pre, ldp=N)
    ref_pre = A.double() @ W.double().t() + bias.double()
    assert_close(pre.float().cpu(), ref_pre.cpu(), 2e-3, 2 ** -8, what="pre-activation")
    assert_close(h.float().cpu(), torch.nn.functional.gelu(ref_pre).cpu(), 3e-3, 2 ** -7, what="gelu")
    y = torch.empty((M, N), device=DEV, dtype=BF)
    vf.gemm(A, W, y, M, N, K, K, K, N, bias=bias, resid=res, ldr=N)
    assert_close(y.float().cpu(), (ref_pre + res.double()).cpu(), 3e-3, 2 ** -8, what="bias + residual")
    dy = _r(M, N, seed=7).to(BF).to(DEV)
    dx = torch.empty((M, K), device=DEV, dtype=BF)
    aux = _r(M, K, seed=8, scale=2.0).to(BF).to(DEV)
    vf.gemm(dy, W, dx, M, K, N, N, K, K, flags=vf.B_KM | vf.GELU_BWD, aux=aux, ldaux=K)
    xa = aux.double().requires_grad_(True)
    torch.nn.functional.gelu(xa).sum().backward()
    assert_close(dx.float().cpu(), ((dy.double() @ W.double()) * xa.grad).cpu(), 3e-3, 2 ** -7, what="dgrad * gelu'")
    # weight gradient: split-K, accumulate, bit-reproducible
    dW = torch.full((N, K), 0.25, device=DEV)
    outs = []
    for _ in range(2):
        dW = torch.full((N, K), 0.25, device=DEV)
        vf.gemm(dy, A, dW, N, K, M, N, K, K, flags=vf.A_KM | vf.B_KM | vf.OUT_F32, ksplit=4, accumulate=True)
        outs.append(dW)
    assert torch.equal(outs[0], outs[1])
    assert_close_scaled(dW.cpu(), (dy.double().t() @ A.double() + 0.25).cpu(), 2e-5, what="split-K weight gradient")
    # two-level batch: (image, head) strided slices of a qkv-like buffer
    Bn, Hn, T, dh = 3, 4, 37, 64
    Dm = Hn * dh
    qkv = _r(Bn, T, 3 * Dm, seed=9).to(BF).to(DEV)
    Tp = 48
    S = torch.zeros((Bn * Hn, T, Tp), device=DEV, dtype=BF)
    vf.gemm(qkv, qkv, S, T, T, dh, 3 * Dm, 3 * Dm, Tp, b_off=Dm, batch=Bn * Hn, batch2=Hn, sA=(T * 3 * Dm, dh),
            sB=(T * 3 * Dm, dh), sC=(Hn * T * Tp, T * Tp), alpha=0.125)
    q = qkv[:, :, :Dm].reshape(Bn, T, Hn, dh).permute(0, 2, 1, 3).double()
    k = qkv[:, :, Dm:2 * Dm].reshape(Bn, T, Hn, dh).permute(0, 2, 1, 3).double()
    ref = 0.125 * q @ k.transpose(-1, -2)
    assert_close(S.view(Bn, Hn, T, Tp)[..., :T].float().cpu(), ref.cpu(), 3e-3, 2 ** -8, what="batched scores")


def test_vit_row_kernels():
    from mclstexp_amd import _lib, vit_fused as vf
    L = _lib.lib()
    rows, D = 333, 768
    x = _r(rows, D, seed=1, scale=2.0).to(BF).to(DEV)
    ln = torch.nn.LayerNorm(D, eps=1e-6).to(DEV)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.uniform_(-0.3, 0.3)
    y, mean, rstd = vf.ln_fwd(x, ln, rows)
    x64 = x.double().requires_grad_(True)
    ln64 = copy.deepcopy(ln).double()
    ref = ln64(x64)
    assert_close(y.float().cpu(), ref.detach().cpu(), 2e-3, 2 ** -8, what="layernorm fwd")
    dy = _r(rows, D, seed=2).to(BF).to(DEV)
    add = _r(rows, D, seed=3).to(BF).to(DEV)
    ref.backward(dy.double())
    dx, dg, db = vf.ln_bwd(dy, x, ln, mean, rstd, add, rows)
    if dg is None:                           # Parameters without .grad get a dense one created and accumulated into
        dg, db = ln.weight.grad, ln.bias.grad
    assert_close(dx.float().cpu(), (x64.grad + add.double()).cpu(), 4e-3, 2 ** -7, what="layernorm dx + add")
    assert_close_scaled(dg.cpu(), ln64.weight.grad.cpu(), 2e-3, what="dgamma")
    assert_close_scaled(db.cpu(), ln64.bias.grad.cpu(), 2e-3, what="dbeta")
    b = torch.nn.Parameter(torch.zeros(2304, device=DEV))
    d2 = _r(rows, 2304, seed=4).to(BF).to(DEV)
    g = vf.bias_grad(d2, b, rows)
    g = b.grad if g is None else g          # a Parameter without .grad gets a dense one created and accumulated into
    assert_close_scaled(g.cpu(), d2.double().sum(0).cpu(), 1e-5, what="bias grad")
    # softmax fwd / bwd in place, padded rows: (197, 208) / (50, 56) take the 16-byte half-wave kernels, odd leading
    # dimensions the scalar ones; the padding columns hold garbage on entry and exact zeros on exit
    for R, n, ld in ((500, 197, 208), (37, 197, 199), (301, 50, 56), (64, 50, 51), (9, 256, 256)):
        s = torch.full((R, ld), 7.0, device=DEV, dtype=BF)
        s[:, :n] = _r(R, n, seed=5, scale=4.0).to(BF).to(DEV)
        s0 = s.clone()
        _lib.check(L.mcl_softmax_bf16_fwd(s.data_ptr(), ld, R, n, vf._st()))
        p64 = torch.softmax(s0[:, :n].double(), dim=1)
        assert_close(s[:, :n].float().cpu(), p64.cpu(), 1e-3, 2 ** -8, what="softmax %s" % ((R, n, ld),))
        assert (s[:, n:] == 0).all()
        dp = torch.full((R, ld), -3.0, device=DEV, dtype=BF)
        dp[:, :n] = _r(R, n, seed=6).to(BF).to(DEV)
        dp0 = dp.clone()
        _lib.check(L.mcl_softmax_bf16_bwd(s.data_ptr(), dp.data_ptr(), ld, R, n, 0.125, vf._st()))
        pp = s[:, :n].double()
        refd = pp * (dp0[:, :n].double() - (pp * dp0[:, :n].double()).sum(1, keepdim=True)) * 0.125
        assert_close(dp[:, :n].float().cpu(), refd.cpu(), 1e-4, 2 ** -7, what="softmax bwd %s" % ((R, n, ld),))
        assert (dp[:, n:] == 0).all()


@pytest.mark.parametrize("name,B,fused_attn", [("vit_base_patch32_224", 6, True), ("vit_base_patch16_224", 3, True),
                                             ("vit_base_patch32_224", 4, False)])
def test_vit_fused_matches_fp64_module(name, B, fused_attn, monkeypatch):
    """Whole encoder forward + backward on the bf16 kernels vs an fp64 run of the same module (timm layout restated in
    backbones.py): features 3e-2 of max|feature|, parameter gradients median relative deviation 3e-2, max 0.25.  With the
    fused attention core (csrc/vit_attention.hip) and, once, with the batched GEMM + softmax path that sequences beyond 224
    tokens still take."""
    from mclstexp_amd import vit_fused as _vf
    from mclstexp_amd.backbones import ImageEncoder_VIT
    from mclstexp_amd.vit_fused import vit_features_fused
    monkeypatch.setattr(_vf, "FUSED_ATTN", fused_attn)
    torch.manual_seed(0)
    enc = ImageEncoder_VIT(name)
    with torch.no_grad():
        for p in enc.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))             # non-trivial biases / LayerNorm parameters
    x = _r(B, 3, 224, 224, seed=5).abs().to(DEV)
    dy = _r(B, 768, seed=6).to(DEV)
    ref64 = copy.deepcopy(enc).double().to(DEV).train()
    y64 = ref64(x.double())
    y64.backward(dy.double())
    fus = copy.deepcopy(enc).to(DEV).train()
    y = vit_features_fused(fus.model, x)
    y.backward(dy)
    scale = y64.abs().max().item()
    err = (y.double() - y64).abs().max().item() / scale
    devs, names = [], []
    for (n, p64), (_, q) in zip(ref64.named_parameters(), fus.named_parameters()):
        assert q.grad is not None, n
        devs.append(((q.grad.double() - p64.grad).abs().max() / (p64.grad.abs().max() + 1e-30)).item())
        names.append(n)
    devs = np.array(devs)
    print(f"{name} B={B}: feature err {err:.2e}; grad rel-dev median {np.median(devs):.2e} max {devs.max():.2e} "
          f"({names[int(devs.argmax())]})")
    assert err <= 3e-2, err
    assert np.median(devs) <= 3e-2 and devs.max() <= 0.25, (np.median(devs), devs.max(), names[int(devs.argmax())])


@pytest.mark.parametrize("name,B", [("vit_base_patch32_224", 5), ("vit_base_patch16_224", 2)])
def test_vit_fp32_matches_fp64_module(name, B):
    """The fp32 ("reference numerics", /root/reference/model.py:104-116) encoder on this library's fp32 kernels -- mcl_gemm in
    exact-fp32 MFMA mode, fp32 LayerNorm / GELU, the fp32 attention core with one sequence per image -- forward + backward vs an
    fp64 run of the same module: 2e-5 of max on the features, 2e-4 on every parameter gradient (fp32 rounding only)."""
    from mclstexp_amd.backbones import ImageEncoder_VIT
    from mclstexp_amd.vit_fused import vit_features_fp32
    torch.manual_seed(0)
    enc = ImageEncoder_VIT(name)
    with torch.no_grad():
        for p in enc.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    x = _r(B, 3, 224, 224, seed=5).abs().to(DEV)
    dy = _r(B, 768, seed=6).to(DEV)
    ref64 = copy.deepcopy(enc).double().to(DEV).train()
    y64 = ref64(x.double())
    y64.backward(dy.double())
    own = copy.deepcopy(enc).to(DEV).train()
    y = vit_features_fp32(own.model, x)
    y.backward(dy)
    err = ((y.double() - y64).abs().max() / y64.abs().max()).item()
    devs, names = [], []
    for (n, p64), (_, q) in zip(ref64.named_parameters(), own.named_parameters()):
        assert q.grad is not None, n
        devs.append(((q.grad.double() - p64.grad).abs().max() / (p64.grad.abs().max() + 1e-30)).item())
        names.append(n)
    devs = np.array(devs)
    print(f"{name} B={B} fp32: feature err {err:.2e}; grad rel-dev median {np.median(devs):.2e} max {devs.max():.2e} "
          f"({names[int(devs.argmax())]})")
    assert err <= 2e-5, err
    assert devs.max() <= 2e-4, (devs.max(), names[int(devs.argmax())])


def test_vit_fp32_selector_and_direct_grads():
    """mclSTExp_Attention('vit', backbone_dtype=None) takes the fp32 own-kernel path in train and eval mode, and with dense
    .grad tensors present (FusedAdam's bucket) the gradients accumulate in place to the same values."""
    from mclstexp_amd.model import mclSTExp_Attention
    torch.manual_seed(0)
    m = mclSTExp_Attention("vit", 1.0, 768, 171, 256, 8, 64, 2, backbone_dtype=None).to(DEV).train()
    x = _r(3, 3, 224, 224, seed=9).abs().to(DEV)
    seed = _r(3, 768, seed=10).to(DEV)
    enc = m.image_encoder if hasattr(m, "image_encoder") else m.image_ecode
    m.encode_image(x).backward(seed)
    g1 = {n: p.grad.clone() for n, p in enc.named_parameters()}
    for p in enc.parameters():
        p.grad.zero_()                                     # dense .grad present: the kernels add straight into it
    m.encode_image(x).backward(seed)
    for n, p in enc.named_parameters():
        assert torch.allclose(p.grad, g1[n], rtol=1e-5, atol=1e-6 * float(g1[n].abs().max()) + 1e-12), n
    m.eval()
    with torch.no_grad():
        y_eval = m.encode_image(x)
    m.train()
    with torch.no_grad():
        y_train = m.encode_image(x)
    assert torch.equal(y_eval, y_train)                    # no dropout / BatchNorm in the encoder: same arithmetic


def test_glue_kernels_vs_torch():
    """csrc/glue.hip: the ViT's token assembly / token mean / embedding gradients, the strided copy, the strided 4-D pack and the
    rotated weight against the torch expressions they replace (bit for bit where the arithmetic is the same)."""
    import ctypes as C
    from mclstexp_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, T, D, p = 3, 5, 16, 8
    np_ = T - 1
    g = torch.Generator().manual_seed(3)
    cls, pos = torch.randn(D, generator=g).to(DEV), torch.randn(T * D, generator=g).to(DEV)
    tok = torch.randn(B * np_, D, generator=g).to(DEV)
    ref = torch.cat([cls.view(1, 1, D).expand(B, 1, D), tok.view(B, np_, D)], dim=1) + pos.view(1, T, D)
    x = torch.empty(B, T, D, device=DEV)
    _lib.check(L.mcl_vit_assemble_f32(tok.data_ptr(), cls.data_ptr(), pos.data_ptr(), x.data_ptr(), B, T, D, st))
    assert torch.equal(x, ref)
    x16 = torch.full((B, T, D), 9.0, device=DEV, dtype=BF)
    _lib.check(L.mcl_vit_cls_row(cls.data_ptr(), pos.data_ptr(), x16.data_ptr(), B, T, D, 1, st))
    assert torch.equal(x16[:, 0], (cls + pos[:D]).to(BF).expand(B, D)) and bool((x16[:, 1:] == 9.0).all())
    for dt, t in ((0, x), (1, ref.to(BF))):
        feat = torch.empty(B, D, device=DEV)
        _lib.check(L.mcl_vit_token_mean_fwd(t.data_ptr(), feat.data_ptr(), B, T, D, dt, st))
        assert_close(feat.cpu(), t[:, 1:].double().mean(dim=1).float().cpu(), 1e-6, 1e-6, what="token mean")
        dfeat = torch.randn(B, D, generator=g).to(DEV)
        dx = torch.empty_like(t)
        _lib.check(L.mcl_vit_token_mean_bwd(dfeat.data_ptr(), dx.data_ptr(), B, T, D, dt, st))
        want = torch.zeros(B, T, D, device=DEV)
        want[:, 1:] = (dfeat / float(np_)).unsqueeze(1)
        assert torch.equal(dx, want.to(t.dtype))
        dtok = torch.empty(B * np_, D, device=DEV, dtype=t.dtype)
        _lib.check(L.mcl_vit_tokens_extract(t.data_ptr(), dtok.data_ptr(), B, T, D, dt, st))
        assert torch.equal(dtok.view(B, np_, D), t[:, 1:])
        dpos, dcls = torch.ones(T * D, device=DEV), torch.ones(D, device=DEV)
        _lib.check(L.mcl_vit_pos_grad(t.data_ptr(), dpos.data_ptr(), dcls.data_ptr(), B, T, D, dt, 1, st))    # += dpos, = dcls
        s = t.float().sum(dim=0).reshape(-1)
        assert_close(dpos.cpu(), (1.0 + s).cpu(), 1e-6, 1e-6, what="dpos (accumulated)")
        assert_close(dcls.cpu(), s[:D].cpu(), 1e-6, 1e-6, what="dcls (overwritten)")
        z = t.clone()
        _lib.check(L.mcl_vit_zero_cls_rows(z.data_ptr(), B, T, D, dt, st))
        assert bool((z[:, 0] == 0).all()) and torch.equal(z[:, 1:], t[:, 1:])
    # patch tokens with a leading zero row, bf16 and fp32, NCHW and channels-last images
    img = torch.rand(B, 3, 2 * p, 2 * p, generator=g).to(DEV)
    K0 = 3 * p * p
    want = img.reshape(B, 3, 2, p, 2, p).permute(0, 2, 4, 1, 3, 5).reshape(B, 4, K0)
    for im in (img, img.contiguous(memory_format=torch.channels_last)):
        for f32 in (0, 1):
            out = torch.full((B, 5, K0), 7.0, device=DEV, dtype=torch.float32 if f32 else BF)
            _lib.check(L.mcl_vit_patchify_tokens(im.data_ptr(), *im.stride(), B, 2 * p, 2 * p, p, out.data_ptr(), 1, f32, st))
            assert bool((out[:, 0] == 0).all()) and torch.equal(out[:, 1:], want.to(out.dtype))
    # strided 4-D pack (a channels-last Conv2d weight -> (D, c, iy, ix) matrix, with the cast) and its accumulating inverse
    w = torch.randn(6, 3, p, p, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    for dt, tdt in ((0, torch.float32), (1, BF)):
        m = torch.empty(6, K0, device=DEV, dtype=tdt)
        _lib.check(L.mcl_strided4_f32(w.data_ptr(), 6, 3, p, p, *w.stride(), m.data_ptr(), K0, p * p, p, 1, dt, 0, st))
        assert torch.equal(m, w.reshape(6, K0).to(tdt))
    gacc = torch.ones_like(w)                                                 # channels-last strides
    gw = torch.randn(6, K0, generator=g).to(DEV)
    _lib.check(L.mcl_strided4_f32(gw.data_ptr(), 6, 3, p, p, K0, p * p, p, 1, gacc.data_ptr(), *gacc.stride(), 0, 1, st))
    assert torch.equal(gacc, 1.0 + gw.view(6, 3, p, p))
    # 2-D row copy: a channel slice of a channels-last buffer -> dense rows (16-, 4- and 2-byte paths)
    for C_, c0, c1, tdt in ((64, 8, 40, torch.float32), (64, 3, 12, torch.float32), (67, 5, 20, BF)):
        buf = torch.randn(2, C_, 5, 7, generator=g).to(DEV).to(tdt).contiguous(memory_format=torch.channels_last)
        sl = buf[:, c0:c1]
        from mclstexp_amd.densenet_fused import dense_cl
        d = dense_cl(sl)
        assert d.is_contiguous(memory_format=torch.channels_last) and torch.equal(d, sl)
    # rotated, role-swapped weight of the backward-data "same" convolution
    Co, k, Ci = 5, 3, 7
    wk = torch.randn(Co, Ci, k, k, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)     # (Co, k, k, Ci) storage
    for dt, tdt in ((0, torch.float32), (1, BF)):
        a = wk.to(tdt).contiguous(memory_format=torch.channels_last)
        wf = torch.empty(Ci, k * k * Co, device=DEV, dtype=tdt)
        _lib.check(L.mcl_weight_rot180(a.data_ptr(), wf.data_ptr(), Co, k, Ci, dt, st))
        ref_wf = a.permute(0, 2, 3, 1).flip(1, 2).permute(3, 1, 2, 0).contiguous().reshape(Ci, k * k * Co)
        assert torch.equal(wf, ref_wf)

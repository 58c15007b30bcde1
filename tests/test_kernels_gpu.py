"""Per-kernel parity: each C-ABI entry point (through ops.py) against the CPU oracle / fp64 torch on
the same seeded inputs.  Run on the MI355X box:  pytest -m gpu."""
import math

import numpy as np
import pytest
import torch

from helpers import assert_close, assert_close_scaled

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.fixture(scope="module")
def ops():
    from mclstexp_amd import _lib, ops as _ops
    _lib.lib()  # must load: no fallback
    _ops.set_compute("f32")
    return _ops


# ------------------------------------------------------------------ GEMM
GEMM_SHAPES = [(128, 1000, 1000), (33, 171, 785), (8, 256, 1024), (1, 64, 64), (65, 129, 33), (128, 1536, 785)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_linear_fwd_and_bwd_layouts(ops, M, N, K):
    x, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3)
    dy = _rand(M, N, seed=4)
    xd, Wd, bd, dyd = x.to(DEV), W.to(DEV), b.to(DEV), dy.to(DEV)
    y, _ = ops.linear_fwd(xd, Wd, bd)
    ref = x.double() @ W.double().t() + b.double()
    assert_close_scaled(y.cpu(), ref, 2e-6, what="y = x W^T + b")
    dx = ops.linear_bwd_data(dyd, Wd)
    assert_close_scaled(dx.cpu(), dy.double() @ W.double(), 2e-6, what="dx = dy W")
    dW = ops.linear_bwd_weight(dyd, xd)
    assert_close_scaled(dW.cpu(), dy.double().t() @ x.double(), 2e-6, what="dW = dy^T x")
    db = ops.colsum(dyd)
    assert_close_scaled(db.cpu(), dy.double().sum(0), 2e-6, what="db")


def test_gemm_epilogues(ops):
    from oracle import ref_cpu
    M, N, K = 37, 171, 300
    x, W, b, r = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.1), _rand(N, seed=3), _rand(M, N, seed=5)
    xd, Wd, bd, rd = x.to(DEV), W.to(DEV), b.to(DEV), r.to(DEV)
    pre_ref = x.double() @ W.double().t() + b.double()
    y, pre = ops.linear_fwd(xd, Wd, bd, gelu=True, save_pre=True, resid=rd)
    assert_close_scaled(pre.cpu(), pre_ref, 2e-6, what="pre-activation")
    assert_close_scaled(y.cpu(), ref_cpu.gelu_erf(pre_ref) + r.double(), 2e-6, what="gelu + resid")
    dy = _rand(M, N, seed=7)
    aux = _rand(M, K, seed=8, scale=2.0)
    dx = ops.linear_bwd_data(dy.to(DEV), Wd, gelu_bwd_aux=aux.to(DEV), resid=xd)
    ref = (dy.double() @ W.double()) * ref_cpu.gelu_erf_grad(aux.double()) + x.double()
    assert_close_scaled(dx.cpu(), ref, 2e-6, what="gelu-bwd epilogue")
    y2, _ = ops.linear_fwd(xd, Wd, None, alpha=0.25)
    assert_close_scaled(y2.cpu(), 0.25 * (x.double() @ W.double().t()), 2e-6, what="alpha, no bias")


def test_gemm_splitk_matches_one_pass_and_is_deterministic(ops):
    """Skinny problems (M = a batch of spots) run as K slices + a fixed-order merge with the epilogue."""
    from mclstexp_amd import _lib
    from oracle import ref_cpu
    L = _lib.lib()
    M, N, K = 128, 1536, 1000
    assert L.mcl_gemm_auto_ksplit(M, N, K, 1) > 1 and L.mcl_gemm_auto_ksplit(4096, 4096, 1000, 1) == 1
    assert L.mcl_gemm_workspace_floats(M, N, 1, 5) == 5 * M * N
    x, W, b, r = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3), _rand(M, N, seed=5)
    xd, Wd, bd, rd = x.to(DEV), W.to(DEV), b.to(DEV), r.to(DEV)
    outs = []
    for split in (True, False, True):
        ops.SPLIT_K = split
        try:
            outs.append(ops.linear_fwd(xd, Wd, bd, gelu=True, save_pre=True, resid=rd))
        finally:
            ops.SPLIT_K = True
    (y_s, pre_s), (y_1, pre_1), (y_s2, pre_s2) = outs
    assert torch.equal(y_s, y_s2) and torch.equal(pre_s, pre_s2)                 # run-to-run bit-identical
    pre_ref = x.double() @ W.double().t() + b.double()
    assert_close_scaled(pre_s.cpu(), pre_ref, 2e-6, what="split-K pre-activation")
    assert_close_scaled(y_s.cpu(), ref_cpu.gelu_erf(pre_ref) + r.double(), 2e-6, what="split-K gelu + resid")
    assert_close_scaled(y_s.cpu(), y_1.cpu().double(), 1e-6, what="split-K vs one pass")
    # ragged K slices and a batched problem (batch * ksplit workgroup layers)
    A3, B3 = _rand(3, 40, 500, seed=6).to(DEV), _rand(3, 500, 70, seed=7).to(DEV)
    C3 = torch.empty(3, 40, 70, device=DEV)
    ops.gemm_raw(40, 70, 500, 3, A3, 500, 1, 40 * 500, B3, 70, 1, 500 * 70, C3, 70, 40 * 70)
    assert L.mcl_gemm_auto_ksplit(40, 70, 500, 3) > 1
    assert_close_scaled(C3.cpu(), A3.cpu().double() @ B3.cpu().double(), 2e-6, what="batched split-K")


@pytest.mark.parametrize("M,N,K,batch", [(128, 1536, 1000, 1), (128, 256, 1024, 1), (33, 171, 785, 1), (40, 70, 500, 3), (128, 1000, 512, 1)])
def test_gemm_splitk_one_launch_equals_two_launches(ops, M, N, K, batch):
    """The last-arriver merge inside the product's launch == partials + a second launch, bit for bit (same slice order), with
    every epilogue; repeated calls find the arrival counters clean."""
    from mclstexp_amd import _lib
    assert _lib.lib().mcl_gemm_auto_ksplit(M, N, K, batch) > 1
    A, B = _rand(batch, M, K, seed=1).to(DEV), _rand(batch, K, N, seed=2, scale=1 / math.sqrt(K)).to(DEV)
    bias, resid, aux = _rand(N, seed=3).to(DEV), _rand(M, N, seed=4).to(DEV), _rand(M, N, seed=5).to(DEV)

    def run(one, **kw):
        ops.SPLIT_K_ONE_LAUNCH = one
        try:
            Cm = torch.full((batch, M, N), 0.5, device=DEV)
            pre = torch.empty(M, N, device=DEV) if kw.pop("pre", False) else None
            ops.gemm_raw(M, N, K, batch, A, K, 1, M * K, B, N, 1, K * N, Cm, N, M * N, pre_out=pre, ldp=N if pre is not None else 0,
                         **kw)
            torch.cuda.synchronize()
            return Cm, pre
        finally:
            ops.SPLIT_K_ONE_LAUNCH = False

    cases = [dict(), dict(flags=_lib.EPI_ACCUM, alpha=0.5)]
    if batch == 1:
        cases += [dict(flags=_lib.EPI_GELU, bias=bias, resid=resid, ldr=N, pre=True),
                  dict(flags=_lib.EPI_GELU_BWD, aux=aux, ldaux=N, resid=resid, ldr=N)]
    for kw in cases:
        two = run(False, **dict(kw))
        for _ in range(3):
            one = run(True, **dict(kw))
            assert torch.equal(one[0], two[0]), kw
            if two[1] is not None:
                assert torch.equal(one[1], two[1]), kw
    assert_close_scaled(run(True)[0].cpu(), A.cpu().double() @ B.cpu().double(), 2e-6, what="one-launch split-K")


def test_linear_bwd_weight_accumulates_into_param_grad(ops):
    """Spot-path weight gradients add straight into an existing fp32 .grad (FusedAdam's flat bucket), also through the
    split-K second pass; without a .grad they are returned."""
    for M, N, K in ((128, 1000, 1000), (128, 40, 1536), (33, 171, 785)):
        dy, x = _rand(M, N, seed=1).to(DEV), _rand(M, K, seed=2).to(DEV)
        w = torch.nn.Parameter(torch.zeros(N, K, device=DEV))
        assert ops.linear_bwd_weight(dy, x, w) is not None            # no .grad yet: returned to autograd
        w.grad = _rand(N, K, seed=3).to(DEV)
        g0 = w.grad.clone()
        assert ops.linear_bwd_weight(dy, x, w) is None
        ref = g0.cpu().double() + dy.cpu().double().t() @ x.cpu().double()
        assert_close_scaled(w.grad.cpu(), ref, 2e-6, what="param.grad += dy^T x (%d, %d, %d)" % (M, N, K))


def test_gemm_unaligned_views(ops):
    """Operands that are strided views (q/k/v slices of qkv) and 4-byte-aligned-only bases."""
    B, H, d = 19, 3, 64
    qkv = _rand(B, 3 * H * d, seed=3).to(DEV)
    out, P = ops.attention_core_fwd_unfused(qkv, H, d)
    q, k, v = qkv.cpu().double().view(B, 3, H, d).permute(1, 2, 0, 3)
    Pref = torch.softmax(q @ k.transpose(1, 2) * d ** -0.5, -1)
    assert_close(P.cpu(), Pref, 2e-6, what="attention probabilities")
    assert_close_scaled((Pref @ v).permute(1, 0, 2).reshape(B, H * d), out.cpu().double(), 4e-6, what="attention out")
    base = _rand(5 * 40 + 1, seed=9).to(DEV)
    x = base[1:].view(5, 40)                      # data_ptr 4-byte aligned only
    W = _rand(7, 40, seed=10).to(DEV)
    y, _ = ops.linear_fwd(x, W)
    assert_close_scaled(y.cpu(), x.cpu().double() @ W.cpu().double().t(), 2e-6, what="unaligned base")


@pytest.mark.parametrize("B,H", [(128, 8), (19, 3), (8, 8), (300, 2), (257, 1), (33, 8)])
def test_attention_fused_fwd_bwd(ops, B, H):
    """csrc/attention.hip (head dimension 64, fp32 on the matrix cores, online softmax over key blocks of 128, no (h, B, B)
    tensor in HBM) vs fp64 autograd of model.py:52-56 on the same qkv: output, log-sum-exp and all three gradients; ragged
    query / key blocks, more than one key block, and agreement with the GEMM + softmax sequence it replaces."""
    d = 64
    qkv = (_rand(B, 3 * H * d, seed=B + H) * 1.5).to(DEV)
    dout = _rand(B, H * d, seed=B + H + 1).to(DEV)
    out, lse = ops.attention_core_fwd(qkv, H, d)
    assert lse.shape == (H, B)
    dqkv = ops.attention_core_bwd(dout, qkv, out, lse, H, d)
    x = qkv.double().clone().requires_grad_(True)
    q, k, v = x.view(B, 3, H, d).permute(1, 2, 0, 3)
    sc = q @ k.transpose(1, 2) * d ** -0.5
    ref = (torch.softmax(sc, -1) @ v).permute(1, 0, 2).reshape(B, H * d)
    ref.backward(dout.double())
    assert_close_scaled(out.cpu(), ref.detach().cpu(), 3e-6, what="fused attention out")
    assert_close(lse.cpu(), torch.logsumexp(sc, -1).detach().cpu(), 1e-5, rtol=1e-6, what="row log-sum-exp")
    assert_close_scaled(dqkv.cpu(), x.grad.cpu(), 5e-6, what="fused attention dqkv")
    out_u, P = ops.attention_core_fwd_unfused(qkv, H, d)
    assert_close_scaled(out.cpu(), out_u.cpu().double(), 3e-6, what="fused vs unfused out")
    assert_close_scaled(dqkv.cpu(), ops.attention_core_bwd(dout, qkv, out_u, P, H, d).cpu().double(), 5e-6, what="fused vs unfused dqkv")


def test_gemm_bf16_mode(ops):
    M, N, K = 128, 256, 1000
    x, W = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K))
    ops.set_compute("bf16")
    try:
        y, _ = ops.linear_fwd(x.to(DEV), W.to(DEV))
        dW = ops.linear_bwd_weight(_rand(M, N, seed=4).to(DEV), x.to(DEV))
    finally:
        ops.set_compute("f32")
    ref = x.bfloat16().double() @ W.bfloat16().double().t()      # operands rounded to bf16, fp32 accumulate
    assert_close_scaled(y.cpu(), ref, 2e-5, what="bf16 MFMA vs bf16-rounded operands")
    ref_w = _rand(M, N, seed=4).bfloat16().double().t() @ x.bfloat16().double()
    assert_close_scaled(dW.cpu(), ref_w, 2e-5, what="bf16 dW")
    # and within bf16 tolerance of the fp32 product
    assert_close_scaled(y.cpu(), x.double() @ W.double().t(), 1e-2, what="bf16 vs fp32")


# ------------------------------------------------------------------ LayerNorm / softmax
@pytest.mark.parametrize("rows,cols", [(128, 1000), (33, 171), (8, 785), (5, 256), (3, 3467)])
def test_layernorm_fwd_bwd(ops, rows, cols):
    from oracle import ref_cpu
    x = _rand(rows, cols, seed=1, scale=3.0) + 0.5
    g, b = 1 + 0.1 * _rand(cols, seed=2), 0.1 * _rand(cols, seed=3)
    dy, add = _rand(rows, cols, seed=4), _rand(rows, cols, seed=5)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yref = ref_cpu.layer_norm(xr, gr, br)
    yref.backward(dy)
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), g.to(DEV), b.to(DEV))
    assert_close(y.cpu(), yref.detach(), 3e-6, what="layernorm y")
    dx, dg, db = ops.layernorm_bwd(dy.to(DEV), x.to(DEV), g.to(DEV), mean, rstd, dx_add=add.to(DEV))
    assert_close_scaled(dx.cpu(), xr.grad + add, 3e-6, what="layernorm dx (+residual)")
    assert_close_scaled(dg.cpu(), gr.grad, 3e-6, what="dgamma")
    assert_close_scaled(db.cpu(), br.grad, 3e-6, what="dbeta")


def test_softmax_rows(ops):
    from mclstexp_amd import _lib
    R, Cn, scale = 70, 133, 0.125
    s = _rand(R, Cn, seed=1, scale=40.0)
    dp = _rand(R, Cn, seed=2)
    sr = s.clone().double().requires_grad_(True)
    pref = torch.softmax(sr * scale, -1)
    pref.backward(dp.double())
    sd = s.to(DEV)
    _lib.check(_lib.lib().mcl_softmax_rows_fwd(sd.data_ptr(), Cn, R, Cn, scale, ops._stream()))
    assert_close(sd.cpu(), pref.detach(), 1e-6, what="softmax")
    dpd = dp.to(DEV)
    _lib.check(_lib.lib().mcl_softmax_rows_bwd(sd.data_ptr(), dpd.data_ptr(), Cn, R, Cn, scale, ops._stream()))
    assert_close_scaled(dpd.cpu(), sr.grad, 3e-6, what="softmax bwd")


# ------------------------------------------------------------------ position embedding
def test_pos_embed_fwd_bwd_dense_and_sparse(ops):
    from oracle import ref_cpu
    B, G, n_rows = 37, 171, 1000
    expr = _rand(B, G, seed=1)
    pos = torch.floor(_rand(B, 2, seed=2).abs() * 12)          # many duplicates
    pos[3, 0] = 7.9                                            # .long() truncation
    xt, yt = _rand(n_rows, G, seed=3), _rand(n_rows, G, seed=4)
    dout = _rand(B, G, seed=5)
    er, xr, yr = (t.clone().requires_grad_(True) for t in (expr, xt, yt))
    ref = ref_cpu.pos_embed_add(er, pos, xr, yr)
    ref.backward(dout)
    ed, xd, yd = (t.to(DEV).requires_grad_(True) for t in (expr, xt, yt))
    out = ops.PosEmbedAddFn.apply(ed, pos.to(DEV), xd, yd, None)
    assert_close(out.detach().cpu(), ref.detach(), 0.0, what="pos_embed_add (bit exact)")
    out.backward(dout.to(DEV))
    assert_close_scaled(xd.grad.cpu(), xr.grad, 2e-6, what="dense x table grad")
    assert_close_scaled(yd.grad.cpu(), yr.grad, 2e-6, what="dense y table grad")
    assert_close(ed.grad.cpu(), er.grad, 0.0, what="d expr")
    sink = {}
    ed2, xd2, yd2 = (t.to(DEV).requires_grad_(True) for t in (expr, xt, yt))
    ops.PosEmbedAddFn.apply(ed2, pos.to(DEV), xd2, yd2, sink).backward(dout.to(DEV))
    assert xd2.grad is None and yd2.grad is None
    rs = ops.embed_rowgrad(sink["dout"], sink["ix"])
    owners = rs.owner_idx.cpu()
    assert sorted(owners[owners >= 0].tolist()) == sorted(set(pos[:, 0].long().tolist()))
    assert_close_scaled(rs.to_dense(n_rows).cpu(), xr.grad, 2e-6, what="row-sparse -> dense")


# ------------------------------------------------------------------ InfoNCE
@pytest.mark.parametrize("B,P,T", [(128, 256, 1.0), (33, 256, 0.5), (8, 64, 1.0), (1, 256, 1.0), (300, 256, 2.0)])
def test_infonce_vs_oracle(ops, B, P, T):
    from oracle import ref_cpu
    # LayerNorm-ed embeddings: row norm ~ sqrt(P) -> logits up to +-100
    es = torch.nn.functional.layer_norm(_rand(B, P, seed=1), (P,))
    ei = torch.nn.functional.layer_norm(_rand(B, P, seed=2) + 0.3 * es, (P,))
    esr, eir = es.clone().double().requires_grad_(True), ei.clone().double().requires_grad_(True)
    s_ref = ref_cpu.logits(esr, eir, T)
    loss_ref = ref_cpu.symmetric_infonce(s_ref)
    loss_ref.backward()
    loss, d_es, d_ei, S = ops.infonce_fwd_bwd(es.to(DEV), ei.to(DEV), T)
    # north_star: 1e-4 abs; the rtol term only matters for |S| > 200 (T=0.5 case), where one fp32 ulp is
    # already 1.5e-5..3e-5 and the fp64 reference is not reachable to 1e-4 by ANY fp32 accumulation order
    assert_close(S.cpu(), s_ref.detach(), 1e-4, 5e-7, what="logits (1e-4 abs, north_star)")
    assert_close(loss.item(), loss_ref.item(), 1e-4, what="loss (1e-4 abs, north_star)")
    # floor: saturated softmax (p - 1 cancellation) leaves gradients of ~1e-7 whose fp32 exp() noise is ~1e-9
    # and rel 1e-4: S - LSE is formed in fp32 at |S| ~ 64..256 (ulp 4e-6..3e-5) before exp(), then p - 1 cancels
    assert_close_scaled(d_es.cpu(), esr.grad, 1e-4, floor=1e-8, what="dE_spot")
    assert_close_scaled(d_ei.cpu(), eir.grad, 1e-4, floor=1e-8, what="dE_img")


def test_infonce_large_properties(ops):
    """BASELINE full size (global batch 1024/2048): size-independent properties instead of a CPU oracle:
    dS rows/cols each sum to 0 (softmax - one-hot), loss invariant under a joint permutation of pairs,
    loss(a*E, T=a) ... and linearity of the gradient GEMMs."""
    B, P = 2048, 256
    es = torch.nn.functional.layer_norm(_rand(B, P, seed=1), (P,)).to(DEV)
    ei = torch.nn.functional.layer_norm(_rand(B, P, seed=2), (P,)).to(DEV)
    loss, d_es, d_ei, S = ops.infonce_fwd_bwd(es, ei, 1.0)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(DEV)
    loss_p, d_es_p, _, _ = ops.infonce_fwd_bwd(es[perm].contiguous(), ei[perm].contiguous(), 1.0)
    assert abs(loss.item() - loss_p.item()) < 1e-4
    assert_close_scaled(d_es_p.cpu(), d_es[perm].cpu(), 1e-5, what="permutation equivariance of dE_spot")
    # sum_i dE_spot[i] . E_spot[i]  ==  sum_ij dS_ij S_ij == sum_j dE_img[j] . E_img[j]
    a = (d_es.double() * es.double()).sum().item()
    b = (d_ei.double() * ei.double()).sum().item()
    assert abs(a - b) < 1e-5 * max(1.0, abs(a))
    ref = torch.logsumexp(S.double(), 1) - torch.diagonal(S.double())
    ref2 = torch.logsumexp(S.double(), 0) - torch.diagonal(S.double())
    assert abs(0.5 * (ref.mean() + ref2.mean()).item() - loss.item()) < 1e-4


# ------------------------------------------------------------------ Adam
def test_adam_flat_and_table(ops):
    from mclstexp_amd import _lib
    from oracle import ref_cpu
    n = 100003
    p, g = _rand(n, seed=1), _rand(n, seed=2, scale=0.01)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, gd, md, vd = (t.to(DEV) for t in (p, g, m, v))
    L = _lib.lib()
    for t in (1, 2, 3):
        ref_cpu.adam_l2_step(p, g, m, v, t)
        _lib.check(L.mcl_adam_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, 1e-4, 0.9, 0.999,
                                   1e-8, 1e-3, 1 - 0.9 ** t, 1 - 0.999 ** t, ops._stream()))
    assert_close(pd.cpu(), p, 2e-7, what="adam p after 3 steps")
    assert_close_scaled(md.cpu(), m, 1e-6, what="exp_avg")
    assert_close_scaled(vd.cpu(), v, 1e-6, what="exp_avg_sq")
    # table form: dense zero gradient + touched rows
    for G in (171, 1000):
        rows = 4096
        tab = _rand(rows, G, seed=3)
        B = 9
        rg = _rand(B, G, seed=4, scale=0.1)
        owner = torch.tensor([5, -1, 4000, 17, -1, 0, 4095, 33, 2], dtype=torch.int32)
        dense = torch.zeros(rows, G)
        for b_, r_ in enumerate(owner.tolist()):
            if r_ >= 0:
                dense[r_] = rg[b_]
        tm, tv = torch.zeros_like(tab), torch.zeros_like(tab)
        td_, tmd, tvd = tab.to(DEV), tm.to(DEV), tv.to(DEV)
        slot = torch.full((rows,), -1, dtype=torch.int32, device=DEV)
        od, rgd = owner.to(DEV), rg.to(DEV)
        ref_cpu.adam_l2_step(tab, dense, tm, tv, 1)
        _lib.check(L.mcl_row_slot_update(slot.data_ptr(), od.data_ptr(), B, 1, ops._stream()))
        _lib.check(L.mcl_adam_table_step(td_.data_ptr(), tmd.data_ptr(), tvd.data_ptr(), rows, G, slot.data_ptr(),
                                         rgd.data_ptr(), G, 1e-4, 0.9, 0.999, 1e-8, 1e-3, 0.1, 0.001, ops._stream()))
        _lib.check(L.mcl_row_slot_update(slot.data_ptr(), od.data_ptr(), B, 0, ops._stream()))
        assert int((slot != -1).sum().item()) == 0
        assert_close(td_.cpu(), tab, 2e-7, what=f"table adam G={G}")
        assert_close_scaled(tmd.cpu(), tm, 1e-6, what="table exp_avg")


# ------------------------------------------------------------------ composite blocks vs oracle autograd
@pytest.mark.parametrize("B,G", [(33, 171), (128, 1000), (8, 785)])
def test_attn_block_and_head_vs_oracle(ops, B, G):
    from mclstexp_amd import synth
    from oracle import ref_cpu
    params = synth.make_params(G, 1024, layers=1, with_tables=False)
    x = _rand(B, G, seed=1, scale=2.0)
    dy = _rand(B, G, seed=2)
    pr = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xr = x.clone().requires_grad_(True)
    yref = ref_cpu.attn_block(xr, pr, "spot_encoder.0.", 8, 64)
    yref.backward(dy)
    pd = {k: v.to(DEV).requires_grad_(True) for k, v in params.items()}
    xd = x.to(DEV).requires_grad_(True)
    q = "spot_encoder.0."
    y = ops.AttnBlockFn.apply(xd, pd[q + "attn.norm.weight"], pd[q + "attn.norm.bias"], pd[q + "attn.fn.to_qkv.weight"],
                              pd[q + "attn.fn.to_out.0.weight"], pd[q + "attn.fn.to_out.0.bias"],
                              pd[q + "ff.norm.weight"], pd[q + "ff.norm.bias"], pd[q + "ff.fn.net.0.weight"],
                              pd[q + "ff.fn.net.0.bias"], pd[q + "ff.fn.net.3.weight"], pd[q + "ff.fn.net.3.bias"], 8, 64)
    assert_close(y.detach().cpu(), yref.detach(), 2e-5, what="attn_block out")
    y.backward(dy.to(DEV))
    assert_close_scaled(xd.grad.cpu(), xr.grad, 1e-5, what="attn_block dx")
    for k in pr:
        if k.startswith(q):
            assert_close_scaled(pd[k].grad.cpu(), pr[k].grad, 2e-5, what="grad " + k)
    # projection head
    de = _rand(B, 256, seed=3)
    xr2 = x.clone().requires_grad_(True)
    eref = ref_cpu.projection_head(xr2, pr, "spot_projection.")
    eref.backward(de)
    xd2 = x.to(DEV).requires_grad_(True)
    h = "spot_projection."
    e = ops.ProjectionHeadFn.apply(xd2, pd[h + "projection.weight"], pd[h + "projection.bias"], pd[h + "fc.weight"],
                                   pd[h + "fc.bias"], pd[h + "layer_norm.weight"], pd[h + "layer_norm.bias"])
    assert_close(e.detach().cpu(), eref.detach(), 1e-5, what="projection head out")
    e.backward(de.to(DEV))
    assert_close_scaled(xd2.grad.cpu(), xr2.grad, 1e-5, what="head dx")
    for k in pr:
        if k.startswith(h):
            assert_close_scaled(pd[k].grad.cpu(), pr[k].grad, 2e-5, what="grad " + k)


@pytest.mark.parametrize("B,G,direct", [(128, 1000, True), (33, 171, False), (8, 785, True)])
def test_attn_block_grouped_gradient_launches_equal_separate_ones(ops, B, G, direct):
    """A spot layer's weight gradients as one mcl_gemm_group launch and its bias gradients as one mcl_colsum_group launch ==
    the seven separate launches, bit for bit (also when they add into the parameters' .grad)."""
    from mclstexp_amd import synth
    params = synth.make_params(G, 1024, layers=1, with_tables=False)
    q = "spot_encoder.0."
    names = [q + n for n in ("attn.norm.weight", "attn.norm.bias", "attn.fn.to_qkv.weight", "attn.fn.to_out.0.weight",
                             "attn.fn.to_out.0.bias", "ff.norm.weight", "ff.norm.bias", "ff.fn.net.0.weight", "ff.fn.net.0.bias",
                             "ff.fn.net.3.weight", "ff.fn.net.3.bias")]
    x, dy = _rand(B, G, seed=1, scale=2.0), _rand(B, G, seed=2)

    def run(grouped):
        ops.GROUP_LAYER_GRADS = grouped
        try:
            pd = [params[n].to(DEV).requires_grad_(True) for n in names]
            if direct:
                for t in pd:
                    t.grad = torch.full_like(t, 0.125)
            xd = x.to(DEV).requires_grad_(True)
            ops.AttnBlockFn.apply(xd, *pd, 8, 64).backward(dy.to(DEV))
            torch.cuda.synchronize()
            return [xd.grad] + [t.grad for t in pd]
        finally:
            ops.GROUP_LAYER_GRADS = True

    a, b = run(True), run(False)
    for n, u, v in zip(["x"] + names, a, b):
        assert torch.equal(u, v), n


# ------------------------------------------------------------------ fused InfoNCE (bf16 MFMA, logits never in HBM)
def _fused_ref(a, b, inv_t, diag_off, lse_b=None):
    """fp64 reference on the bf16-rounded operands: (lse, diag, dA/coef given lse_b)."""
    a64, b64 = a.bfloat16().double(), b.bfloat16().double()
    S = a64 @ b64.t() * inv_t
    lse = torch.logsumexp(S, dim=1)
    R, C = S.shape
    idx = torch.arange(R)
    ok = (idx + diag_off >= 0) & (idx + diag_off < C)
    diag = torch.zeros(R, dtype=torch.float64)
    diag[ok] = S[idx[ok], idx[ok] + diag_off]
    dA = None
    if lse_b is not None:
        w = torch.exp(S - lse[:, None]) + torch.exp(S - lse_b.double()[None, :])
        w[idx[ok], idx[ok] + diag_off] -= 2.0
        dA = w @ b64
    return lse, diag, dA


@pytest.mark.parametrize("R,C,T,doff", [(128, 128, 1.0, 0), (33, 33, 0.5, 0), (256, 1024, 1.0, 512), (200, 333, 2.0, 100),
                                        (1, 1, 1.0, 0), (512, 512, 1.0, 0), (128, 2048, 1.0, 1920)])
def test_infonce_fused_strip(ops, R, C, T, doff):
    """One orientation of the fused kernel (rows = own embeddings, columns = the other side's), including
    ragged tiles, the split over column ranges and a data-parallel diagonal offset."""
    P = 256
    g = torch.Generator().manual_seed(R * 7 + C)
    a = torch.nn.functional.layer_norm(torch.randn(R, P, generator=g), (P,))
    b = torch.nn.functional.layer_norm(torch.randn(C, P, generator=g), (P,))
    n_pos = max(0, min(R, C - doff))
    b[doff: doff + n_pos] += 0.08 * a[:n_pos]                # positive pairs correlate (not saturated)
    inv_t = 1.0 / T
    lse_b = torch.logsumexp((b.bfloat16().double() @ a.bfloat16().double().t()) * inv_t, dim=1).float()
    lse_ref, diag_ref, dA_ref = _fused_ref(a, b, inv_t, doff, lse_b)
    a16, b16 = ops.cast_bf16(a.to(DEV)), ops.cast_bf16(b.to(DEV))
    assert torch.equal(a16.cpu(), a.bfloat16()), "cast kernel is round-to-nearest-even"
    lse, diag = ops.infonce_fused_lse(a16, b16, inv_t, doff)
    # logits reach +-(16*16)/T; fp32 accumulation of exact bf16 products
    assert_close(lse.cpu(), lse_ref, 2e-4 * max(1.0, inv_t), what="fused lse")
    idx = torch.arange(R)
    ok = (idx + doff < C)
    assert_close(diag.cpu()[ok], diag_ref[ok], 2e-4 * max(1.0, inv_t), what="fused diag")
    dA = ops.infonce_fused_grad(a16, b16, inv_t, lse_ref.float().to(DEV), lse_b.to(DEV), 1.0, doff)
    # w is rounded to bf16 before the second contraction: 2^-9 relative per term
    # + an absolute floor: exponent arguments carry ~1e-5 of fp32 rounding at |logit| ~ 100
    assert_close(dA.cpu(), dA_ref, 6e-3 * float(dA_ref.abs().max()) + 2e-3, what="fused dA")


def test_infonce_fused_matches_unfused_loss_and_grads(ops):
    """Whole symmetric loss: fused bf16 path vs the exact-fp32 unfused kernels on bf16-representable inputs."""
    B, P = 384, 256
    g = torch.Generator().manual_seed(5)
    es = torch.nn.functional.layer_norm(torch.randn(B, P, generator=g), (P,)).bfloat16().float()
    ei = torch.nn.functional.layer_norm(torch.randn(B, P, generator=g) + 0.08 * es, (P,)).bfloat16().float()
    loss_u, des_u, dei_u, _ = ops.infonce_fwd_bwd(es.to(DEV), ei.to(DEV), 1.0)
    loss_f, des_f, dei_f, _ = ops.infonce_fused_fwd_bwd(es.to(DEV), ei.to(DEV), 1.0, min_fused_batch=0)
    assert abs(loss_f.item() - loss_u.item()) < 2e-4 * max(1.0, abs(loss_u.item()))
    assert_close_scaled(des_f.cpu(), des_u.cpu(), 6e-3, what="dE_spot fused vs exact")
    assert_close_scaled(dei_f.cpu(), dei_u.cpu(), 6e-3, what="dE_img fused vs exact")


def test_infonce_fused_spike_forces_rescale(ops):
    """One column far above the running maximum in a LATER tile: the online (max, sum) rescale must fire."""
    R, C, P = 64, 640, 256
    g = torch.Generator().manual_seed(11)
    a = torch.randn(R, P, generator=g) * 0.1
    b = torch.randn(C, P, generator=g) * 0.1
    b[600] = a[7] * 300.0          # huge logit for row 7 in the last tile
    lse_ref, _, _ = _fused_ref(a, b, 1.0, 0)
    lse, _ = ops.infonce_fused_lse(ops.cast_bf16(a.to(DEV)), ops.cast_bf16(b.to(DEV)), 1.0, 0)
    assert_close(lse.cpu(), lse_ref, 1e-3, rtol=1e-5, what="lse with a late spike")


# ------------------------------------------------------------------ fp8 similarity contraction (BASELINE configs[4])
def _quant_ref(x):
    """The quantiser's contract restated with torch's own OCP e4m3 cast: per row the smallest power-of-two scale 2^e
    with max|x| <= 448 * 2^e, q = e4m3_rne(x * 2^-e); returns (bytes, e, dequantised fp32)."""
    amax = x.abs().amax(dim=1)
    e = torch.ceil(torch.log2(amax.double() / 448.0)).clamp(-126, 126)
    e = torch.where(amax > 0, e, torch.zeros_like(e))
    inv = torch.pow(2.0, -e).float()[:, None]
    q = (x * inv).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), e.to(torch.int64), q.float() * torch.pow(2.0, e).float()[:, None]


@pytest.mark.parametrize("rows", [1, 5, 128, 1000])
def test_quant_e4m3_rows(ops, rows):
    """Row-scaled e4m3 quantiser: bytes and E8M0 scale bytes equal torch's float8_e4m3fn cast of the scaled rows, the
    bf16 dequantised copy is exact, dequant(quant) round-trips bit for bit."""
    g = torch.Generator().manual_seed(rows)
    x = torch.nn.functional.layer_norm(torch.randn(rows, 256, generator=g), (256,)) * (0.01 + 3.0 * torch.rand(rows, 1, generator=g))
    if rows > 2:
        x[1] = 0.0                                         # all-zero row: e = 0, q = 0
        x[2, 7] = 448.0 * 4.0                              # exactly on a scale boundary
    packed, deq = ops.quant_e4m3(x.to(DEV))
    qb, e, dq = _quant_ref(x)
    assert torch.equal(packed[:, :256].cpu(), qb), "e4m3 bytes"
    assert torch.equal(packed[:, 256].cpu().to(torch.int64), e + 127), "E8M0 scale bytes"
    assert torch.equal(deq.float().cpu(), dq), "dequantised copy (exact in bf16)"
    assert torch.equal(ops.dequant_e4m3(packed).cpu(), deq.cpu())
    # round-to-nearest error: half a unit in the last of e4m3's 4 significant bits, or half a subnormal step (2^-10 2^e)
    assert ((dq - x).abs() <= torch.maximum(2.0 ** -4 * x.abs(), 2.0 ** -10 * torch.pow(2.0, e).float()[:, None])).all()


@pytest.mark.parametrize("R,C,T", [(128, 128, 1.0), (33, 33, 0.5), (256, 1024, 1.0), (200, 333, 2.0), (1, 1, 1.0),
                                   (129, 2049, 1.0), (2048, 2048, 1.0)])
def test_infonce_fp8_lse(ops, R, C, T):
    """fp8 MFMA row LSE (hardware E8M0 block scales, logits never in HBM) vs fp64 logsumexp of the DEQUANTISED operands:
    2e-3/T absolute.  The products are exact, but v_mfma_scale_f32_32x32x64_f8f6f4 sums its 64 products per instruction
    with less internal precision than an fp32 FMA chain: measured 9e-4 at |S| ~ 100 (30 fp32 ulps), against 1e-4 for
    the bf16 MFMA on the same operands.  Ragged tiles, column splits."""
    g = torch.Generator().manual_seed(R * 7 + C)
    a = torch.nn.functional.layer_norm(torch.randn(R, 256, generator=g), (256,))
    b = torch.nn.functional.layer_norm(torch.randn(C, 256, generator=g), (256,))
    n = min(R, C)
    b[:n] += 0.08 * a[:n]
    inv_t = 1.0 / T
    a8, a16 = ops.quant_e4m3(a.to(DEV))
    b8, b16 = ops.quant_e4m3(b.to(DEV))
    ref = torch.logsumexp((a16.double().cpu() @ b16.double().cpu().t()) * inv_t, dim=1)
    lse = ops.infonce_fp8_lse(a8, b8, inv_t)
    assert_close(lse.cpu(), ref, 2e-3 * max(1.0, inv_t), what="fp8 lse")
    # the bf16 statistics kernel on the dequantised copies sees the same operands
    lse16, diag16 = ops.infonce_fused_lse(a16, b16, inv_t, 0)
    assert_close(lse.cpu(), lse16.cpu(), 2e-3 * max(1.0, inv_t), what="fp8 lse vs bf16 kernel on the dequantised copy")
    diag = ops.infonce_rowdot(a16, b16, inv_t, 0)
    assert_close(diag.cpu()[:n], diag16.cpu()[:n], 1e-4 * max(1.0, inv_t), what="rowdot")


def test_infonce_fp8_loss_and_grads(ops):
    """Whole symmetric loss in fp8 mode (configs[4] global batch 2048): (1) against the closed form in fp64 on the
    DEQUANTISED embeddings -- loss 1e-3 (the fp8 MFMA's internal accumulation, see test_infonce_fp8_lse), gradients
    6e-3 of their maximum (bf16 weights in the second contraction);
    (2) the honest distance from the fp32 loss on the unquantised embeddings: e4m3 keeps 4 significant bits, logits
    (|S| up to ~90 at T = 1) move by ~0.5, the loss by a few percent -- printed and bounded at 5 %."""
    from oracle import ref_cpu
    B = 2048
    g = torch.Generator().manual_seed(17)
    es = torch.nn.functional.layer_norm(torch.randn(B, 256, generator=g), (256,))
    ei = torch.nn.functional.layer_norm(torch.randn(B, 256, generator=g) + 0.15 * es, (256,))
    loss, d_es, d_ei, _ = ops.infonce_fp8_fwd_bwd(es.to(DEV), ei.to(DEV), 1.0)
    _, s16 = ops.quant_e4m3(es.to(DEV))
    _, i16 = ops.quant_e4m3(ei.to(DEV))
    a = s16.double().cpu().requires_grad_(True)
    b = i16.double().cpu().requires_grad_(True)
    ref = ref_cpu.symmetric_infonce(ref_cpu.logits(a, b, 1.0))
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-3 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    assert_close(d_es.cpu(), a.grad, 6e-3 * float(a.grad.abs().max()) + 2e-3 / B, what="dE_spot (fp8 mode)")
    assert_close(d_ei.cpu(), b.grad, 6e-3 * float(b.grad.abs().max()) + 2e-3 / B, what="dE_img (fp8 mode)")
    full = ref_cpu.symmetric_infonce(ref_cpu.logits(es.double(), ei.double(), 1.0)).item()
    print(f"fp8 InfoNCE B={B}: loss {loss.item():.4f} vs fp32-operand loss {full:.4f} "
          f"(rel {abs(loss.item() - full) / max(1.0, abs(full)):.2e})")
    assert abs(loss.item() - full) <= 5e-2 * max(1.0, abs(full))


def test_model_fp8_infonce_mode():
    """mclSTExp_Attention(infonce='fp8'): one training step runs through the fp8 path and lands near the exact mode."""
    from mclstexp_amd import synth
    from mclstexp_amd.model import mclSTExp_Attention
    G, D, B = 171, 1024, 64
    losses = {}
    for mode in ("exact", "fp8"):
        m = mclSTExp_Attention("identity", 1.0, D, G, 256, 8, 64, 2, infonce=mode)
        m.load_state_dict(synth.make_params(G, D, seed=0))
        m.to(DEV).train()
        batch = {k: v.to(DEV) for k, v in synth.make_batch(B, G, image_dim=D, seed=0).items()}
        loss = m(batch)
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
        losses[mode] = loss.item()
    assert abs(losses["fp8"] - losses["exact"]) <= 5e-2 * max(1.0, abs(losses["exact"])), losses


# ------------------------------------------------------------------ BLEEP soft-target CLIP loss
@pytest.mark.parametrize("name", ["clip_b8", "clip_b33_t07", "vit_b16_t05"])
def test_soft_clip_loss_against_reference_fixture(ops, name):
    """ops.soft_clip_fwd_bwd (fp32 MFMA GEMMs + softmax / LSE kernels, closed-form backward through the soft targets)
    against the loss and autograd gradients of the reference's own code (tests/golden/bleep_loss.npz): 1e-4 absolute
    on the loss (north_star's bar for loss/logits), gradients 2e-5 of their largest entry."""
    import os
    from helpers import GOLDEN_DIR, bleep_embeddings
    z = np.load(os.path.join(GOLDEN_DIR, "bleep_loss.npz"))
    B, seed, vit = [int(v) for v in z[name + ".meta"]]
    es, ei = bleep_embeddings(B, seed)
    loss, d_es, d_ei = ops.soft_clip_fwd_bwd(es.to(DEV), ei.to(DEV), float(z[name + ".T"]), bool(vit))
    assert_close(loss.item(), z[name + ".loss"], 1e-4, what="loss (1e-4 abs)")
    assert_close_scaled(d_es.cpu(), z[name + ".d_es"], 2e-5, what="d_es")
    assert_close_scaled(d_ei.cpu(), z[name + ".d_ei"], 2e-5, what="d_ei")
    # autograd wrapper + model hook
    from mclstexp_amd.model import mclSTExp_MLP
    a, b = es.to(DEV).requires_grad_(True), ei.to(DEV).requires_grad_(True)
    (ops.SoftClipLossFn.apply(a, b, float(z[name + ".T"]), bool(vit)) * 2.0).backward()
    assert_close_scaled(a.grad.cpu(), 2.0 * z[name + ".d_es"], 2e-5, what="autograd d_es")
    m = mclSTExp_MLP(float(z[name + ".T"]), 1024, 171, 256, encoder_name="identity")
    m.loss_kind = "bleep_vit" if vit else "bleep"
    assert abs(m._loss(es.to(DEV), ei.to(DEV)).item() - float(z[name + ".loss"])) < 1e-4


def test_soft_clip_loss_large_batch_vs_oracle(ops):
    from oracle import ref_cpu
    g = torch.Generator().manual_seed(5)
    es = torch.nn.functional.layer_norm(torch.randn(1000, 256, generator=g), (256,)) * 0.05
    ei = (0.5 * es + 0.05 * torch.randn(1000, 256, generator=g)).contiguous()
    a, b = es.double().requires_grad_(True), ei.double().requires_grad_(True)
    ref = ref_cpu.bleep_soft_clip_loss(a, b, 0.8)
    ref.backward()
    loss, d_es, d_ei = ops.soft_clip_fwd_bwd(es.to(DEV), ei.to(DEV), 0.8)
    assert_close(loss.item(), ref.item(), 1e-4, what="loss")
    assert_close_scaled(d_es.cpu(), a.grad, 5e-5, what="d_es")
    assert_close_scaled(d_ei.cpu(), b.grad, 5e-5, what="d_ei")


def test_soft_clip_loss_launches_own_kernels_only(ops):
    """the loss was four own GEMMs stitched by a dozen ATen elementwise launches; now every launch
    between the embeddings and the gradients is this library's (csrc/soft_clip.hip for the elementwise middle)."""
    from mclstexp_amd import kernel_audit
    g = torch.Generator().manual_seed(1)
    es = (torch.randn(256, 256, generator=g) * 0.05).to(DEV)
    ei = (torch.randn(256, 256, generator=g) * 0.05).to(DEV)
    ops.soft_clip_fwd_bwd(es, ei, 0.8)
    ks = kernel_audit.step_kernels(lambda: ops.soft_clip_fwd_bwd(es, ei, 0.8))
    bad = kernel_audit.foreign(ks)
    assert not bad, bad
    assert any("soft_clip_mid_kernel" in k for k in ks) and any("symmetrize_kernel" in k for k in ks)


def test_out_of_range_position_raises_like_nn_embedding(ops):
    """positions outside the tables are clamped by the kernel but must not train the wrong rows silently:
    the device flag turns into nn.Embedding's IndexError at the next check; integer position tensors are accepted."""
    G = 16
    xt = torch.zeros(65536, G, device=DEV)
    yt = torch.zeros(65536, G, device=DEV)
    expr = torch.ones(3, G, device=DEV)
    ops.check_position_errors()
    out = ops.PosEmbedAddFn.apply(expr, torch.tensor([[1, 2], [3, 4], [5, 6]], device=DEV), xt, yt, None)   # int64 positions
    assert torch.equal(out, expr)
    ops.check_position_errors()                                   # in range: nothing raised
    ops.PosEmbedAddFn.apply(expr, torch.tensor([[1.0, 2.0], [70000.0, 4.0], [5.0, -3.0]], device=DEV), xt, yt, None)
    with pytest.raises(IndexError):
        ops.check_position_errors()
    ops.check_position_errors()                                   # the flag was cleared

"""The single-launch ProjectionHead (csrc/proj_head.hip, reference model.py:151-168) and the grouped GEMM launch
(mcl_gemm_group) against fp64 torch, against the separate launches, and against themselves (bit-reproducibility, clean arrival
counters over repeated calls).  Run on the MI355X box:  pytest -m gpu."""
import math

import pytest
import torch

from helpers import assert_close_scaled

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


def _head_params(D, seed=0):
    return dict(wp=_rand(256, D, seed=seed + 1, scale=1 / math.sqrt(D)), bp=_rand(256, seed=seed + 2, scale=0.1),
                wf=_rand(256, 256, seed=seed + 3, scale=1 / 16), bf=_rand(256, seed=seed + 4, scale=0.1),
                g=1 + _rand(256, seed=seed + 5, scale=0.2), be=_rand(256, seed=seed + 6, scale=0.1))


def _ref64(x, q, de):
    """fp64 autograd of model.py:160-166."""
    t = {k: v.double().requires_grad_(True) for k, v in q.items()}
    x64 = x.double().requires_grad_(True)
    p = x64 @ t["wp"].t() + t["bp"]
    z = torch.nn.functional.gelu(p) @ t["wf"].t() + t["bf"] + p
    e = torch.nn.functional.layer_norm(z, (256,), t["g"], t["be"], 1e-5)
    e.backward(de.double())
    return e.detach(), x64.grad, {k: v.grad for k, v in t.items()}


def _run(ops, x, q, de, direct=False):
    qd = {k: v.to(DEV).requires_grad_(True) for k, v in q.items()}
    if direct:      # the parameters own dense fp32 .grad buffers (FusedAdam's flat bucket): the kernels add into them
        for v in qd.values():
            v.grad = torch.full_like(v, 0.25)
    xd = x.to(DEV).requires_grad_(True)
    e = ops.ProjectionHeadFn.apply(xd, qd["wp"], qd["bp"], qd["wf"], qd["bf"], qd["g"], qd["be"])
    e.backward(de.to(DEV))
    torch.cuda.synchronize()
    return e.detach().cpu(), xd.grad.cpu(), {k: v.grad.cpu() for k, v in qd.items()}


HEAD_SHAPES = [(128, 1000), (128, 1024), (33, 171), (8, 785), (5, 1024), (1, 64), (300, 1000), (256, 3467), (17, 33)]


@pytest.mark.parametrize("M,D", HEAD_SHAPES)
def test_fused_head_vs_fp64(ops, M, D):
    assert ops.FUSED_HEAD
    x, q, de = _rand(M, D, seed=11, scale=2.0), _head_params(D), _rand(M, 256, seed=12)
    e, dx, gr = _run(ops, x, q, de)
    eref, dxref, gref = _ref64(x, q, de)
    assert_close_scaled(e, eref, 2e-6, what="head out")
    assert_close_scaled(dx, dxref, 3e-6, what="head dx")
    for k in gr:
        assert_close_scaled(gr[k], gref[k], 5e-6, what="head grad " + k)


@pytest.mark.parametrize("M,D", [(128, 1000), (33, 171), (8, 785)])
def test_fused_head_adds_into_parameter_grads_and_repeats_bit_for_bit(ops, M, D):
    x, q, de = _rand(M, D, seed=21, scale=2.0), _head_params(D, seed=7), _rand(M, 256, seed=22)
    _, _, gref = _ref64(x, q, de)
    first = None
    for it in range(4):        # (repeated calls: every call must leave the arrival counters clean for the next)
        e, dx, gr = _run(ops, x, q, de, direct=True)
        for k in gr:
            assert_close_scaled(gr[k] - 0.25, gref[k], 5e-6, what="direct grad " + k)
        if first is None:
            first = (e, dx, gr)
        else:
            assert torch.equal(e, first[0]) and torch.equal(dx, first[1]), "not bit-reproducible"
            for k in gr:
                assert torch.equal(gr[k], first[2][k]), "not bit-reproducible: " + k


def test_fused_head_matches_the_separate_launches(ops, monkeypatch):
    M, D = 128, 1000
    x, q, de = _rand(M, D, seed=31, scale=2.0), _head_params(D, seed=3), _rand(M, 256, seed=32)
    a = _run(ops, x, q, de)
    monkeypatch.setattr(ops, "FUSED_HEAD", False)
    b = _run(ops, x, q, de)
    assert_close_scaled(a[0], b[0], 2e-6, what="fused vs separate: out")
    assert_close_scaled(a[1], b[1], 3e-6, what="fused vs separate: dx")
    for k in a[2]:
        assert_close_scaled(a[2][k], b[2][k], 5e-6, what="fused vs separate: " + k)


@pytest.mark.parametrize("ks", [1, 2, 5, 16])
def test_fused_head_every_slice_count(ops, ks, monkeypatch):
    monkeypatch.setenv("MCL_HEAD_KSPLIT", str(ks))
    M, D = 40, 1000
    x, q, de = _rand(M, D, seed=41, scale=2.0), _head_params(D, seed=5), _rand(M, 256, seed=42)
    e, dx, gr = _run(ops, x, q, de)
    eref, dxref, gref = _ref64(x, q, de)
    assert_close_scaled(e, eref, 2e-6, what=f"head out, {ks} slices")
    assert_close_scaled(gr["wp"], gref["wp"], 5e-6, what=f"head dWp, {ks} slices")


def test_fused_head_under_graph_replay_and_two_streams(ops):
    """Two heads on two streams at once (the step's lanes), captured and replayed: per-weight counters, nothing shared."""
    M = 128
    xs = [_rand(M, 1000, seed=51, scale=2.0).to(DEV), _rand(M, 1024, seed=52, scale=2.0).to(DEV)]
    qs = [{k: v.to(DEV) for k, v in _head_params(1000, seed=1).items()}, {k: v.to(DEV) for k, v in _head_params(1024, seed=2).items()}]
    refs = [ops.proj_head_fwd(x, q["wp"], q["bp"], q["wf"], q["bf"], q["g"], q["be"])[0].clone() for x, q in zip(xs, qs)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    outs = [None, None]
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        outs[0] = ops.proj_head_fwd(xs[0], qs[0]["wp"], qs[0]["bp"], qs[0]["wf"], qs[0]["bf"], qs[0]["g"], qs[0]["be"])[0]
        with torch.cuda.stream(side):
            outs[1] = ops.proj_head_fwd(xs[1], qs[1]["wp"], qs[1]["bp"], qs[1]["wf"], qs[1]["bf"], qs[1]["g"], qs[1]["be"])[0]
        main.wait_stream(side)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)


def test_gemm_group_equals_separate_launches(ops):
    """Three problems of three operand layouts in one launch == three mcl_gemm calls, bit for bit (incl. += into C)."""
    import ctypes as C
    from mclstexp_amd import _lib
    M, N, D = 128, 256, 1000
    dz, a, x, wp = (_rand(M, N, seed=1).to(DEV), _rand(M, N, seed=2).to(DEV), _rand(M, D, seed=3).to(DEV),
                    _rand(N, D, seed=4, scale=0.03).to(DEV))
    base = _rand(N, N, seed=5).to(DEV)

    def problems(c1, c2, c3):
        return [dict(M=N, N=N, K=M, A=dz.data_ptr(), sAm=1, sAk=N, B=a.data_ptr(), sBk=N, sBn=1, C=c1.data_ptr(), ldc=N,
                     flags=_lib.EPI_ACCUM),
                dict(M=N, N=D, K=M, A=dz.data_ptr(), sAm=1, sAk=N, B=x.data_ptr(), sBk=D, sBn=1, C=c2.data_ptr(), ldc=D),
                dict(M=M, N=D, K=N, A=dz.data_ptr(), sAm=N, sAk=1, B=wp.data_ptr(), sBk=D, sBn=1, C=c3.data_ptr(), ldc=D)]

    g1, g2, g3 = base.clone(), torch.empty(N, D, device=DEV), torch.empty(M, D, device=DEV)
    ops.gemm_group(problems(g1, g2, g3))
    s1, s2, s3 = base.clone(), torch.empty(N, D, device=DEV), torch.empty(M, D, device=DEV)
    for kw in problems(s1, s2, s3):
        arg = _lib.gemm_args(batch=1, alpha=1.0, compute=_lib.COMPUTE_F32, **kw)
        _lib.check(_lib.lib().mcl_gemm(C.byref(arg), None), "mcl_gemm")
    torch.cuda.synchronize()
    assert torch.equal(g1, s1) and torch.equal(g2, s2) and torch.equal(g3, s3)
    assert_close_scaled(g2.cpu(), dz.cpu().double().t() @ x.cpu().double(), 2e-6, what="grouped dW")
    # argument checks: more than four problems, a split-K problem
    with pytest.raises(RuntimeError):
        ops.gemm_group(problems(g1, g2, g3) + problems(g1, g2, g3)[:2])
    bad = problems(g1, g2, g3)
    bad[0]["ksplit"] = 2
    bad[0]["workspace"] = g2.data_ptr()
    with pytest.raises(RuntimeError):
        ops.gemm_group(bad)


def test_fused_head_stress_under_traffic(ops):
    """The ticket hand-off (write-through partials, drain, relaxed ticket, last arriver) a few hundred times with fresh data while a
    second stream keeps the memory system busy: every result equals the first-principles value within fp32 noise AND repeats bit for
    bit on an immediate second call."""
    M, D = 128, 1000
    q = {k: v.to(DEV) for k, v in _head_params(D, seed=9).items()}
    noise_stream = torch.cuda.Stream()
    big = torch.empty(64 << 20, device=DEV)
    w64 = {k: v.double() for k, v in q.items()}
    for it in range(200):
        x = _rand(M, D, seed=1000 + it, scale=2.0).to(DEV)
        with torch.cuda.stream(noise_stream):
            big.mul_(1.0001)                                   # 256 MB of traffic beside the head kernels
        e1 = ops.proj_head_fwd(x, q["wp"], q["bp"], q["wf"], q["bf"], q["g"], q["be"])[0]
        e2 = ops.proj_head_fwd(x, q["wp"], q["bp"], q["wf"], q["bf"], q["g"], q["be"])[0]
        assert torch.equal(e1, e2), f"iteration {it}: not reproducible"
        if it % 20 == 0:
            p = x.double() @ w64["wp"].t() + w64["bp"]
            z = torch.nn.functional.gelu(p) @ w64["wf"].t() + w64["bf"] + p
            ref = torch.nn.functional.layer_norm(z, (256,), w64["g"], w64["be"], 1e-5)
            assert float((e1.double() - ref).abs().max()) < 2e-5, f"iteration {it}"
    torch.cuda.synchronize()

"""Data-parallel logic on CPU with gloo, world_size 2 (and 4): the collective plumbing of
mclstexp_amd/dist.py with oracle-backed arithmetic primitives (the HIP primitives need a GPU).
Checks: global InfoNCE == single-process loss on the concatenated embeddings, gradients included;
table-row exchange; flat-bucket gradient all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

from oracle import ref_cpu


class OraclePrims:
    """CPU restatement of dist.HipPrims' five primitives (tests only)."""

    @staticmethod
    def logits(a, b, inv_t):
        return (a @ b.t()) * inv_t

    @staticmethod
    def row_lse(S):
        return torch.logsumexp(S, dim=1)

    @staticmethod
    def col_lse(S):
        return torch.logsumexp(S, dim=0)

    @staticmethod
    def dlogits(S, row_lse, col_lse, row0, col0, coef):
        i = torch.arange(S.shape[0]).unsqueeze(1) + row0
        j = torch.arange(S.shape[1]).unsqueeze(0) + col0
        return coef * (torch.exp(S - row_lse.unsqueeze(1)) + torch.exp(S - col_lse.unsqueeze(0)) - 2.0 * (i == j))

    @staticmethod
    def mm_nn(dS, e):
        return dS @ e

    @staticmethod
    def mm_tn(dS, e):
        return dS.t() @ e


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        td.destroy_process_group()


def _spawn(fn, world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _infonce_case(rank, world):
    from mclstexp_amd import dist as mdist
    b_loc, P, T = 6, 32, 0.7
    g = torch.Generator().manual_seed(1234)
    es_all = torch.nn.functional.layer_norm(torch.randn(world * b_loc, P, generator=g), (P,))
    ei_all = torch.nn.functional.layer_norm(torch.randn(world * b_loc, P, generator=g), (P,))
    sl = slice(rank * b_loc, (rank + 1) * b_loc)
    loss, d_es, d_ei, s_rows = mdist.dist_infonce_fwd_bwd(es_all[sl].clone(), ei_all[sl].clone(), T, td.group.WORLD,
                                                          prims=OraclePrims)
    return loss.item(), d_es, d_ei, s_rows


@pytest.mark.parametrize("world", [2, 4])
def test_dist_infonce_equals_single_process(world):
    outs = _spawn(_infonce_case, world)
    b_loc, P, T = 6, 32, 0.7
    g = torch.Generator().manual_seed(1234)
    es = torch.nn.functional.layer_norm(torch.randn(world * b_loc, P, generator=g), (P,)).requires_grad_(True)
    ei = torch.nn.functional.layer_norm(torch.randn(world * b_loc, P, generator=g), (P,)).requires_grad_(True)
    s = ref_cpu.logits(es, ei, T)
    loss = ref_cpu.symmetric_infonce(s)
    loss.backward()
    for r, (l, d_es, d_ei, s_rows) in enumerate(outs):
        sl = slice(r * b_loc, (r + 1) * b_loc)
        assert abs(l - loss.item()) < 1e-5                      # identical global loss on every rank
        assert torch.allclose(s_rows, s.detach()[sl], atol=1e-5)
        # dE of the GLOBAL mean loss: weight grads are then SUMMED over ranks
        assert torch.allclose(d_es, es.grad[sl], atol=1e-6), (d_es - es.grad[sl]).abs().max()
        assert torch.allclose(d_ei, ei.grad[sl], atol=1e-6)


class OracleFusedPrims:
    """CPU restatement of dist.HipFusedPrims (tests only): bf16-rounded operands, fp64 arithmetic."""

    @staticmethod
    def cast(x):
        return x.bfloat16()

    @staticmethod
    def lse(a16, b16, inv_t, diag_off):
        S = (a16.double() @ b16.double().t()) * inv_t
        idx = torch.arange(S.shape[0])
        return torch.logsumexp(S, dim=1).float(), S[idx, idx + diag_off].float()

    @staticmethod
    def grad(a16, b16, inv_t, lse_a, lse_b, coef, diag_off):
        S = (a16.double() @ b16.double().t()) * inv_t
        w = torch.exp(S - lse_a.double()[:, None]) + torch.exp(S - lse_b.double()[None, :])
        idx = torch.arange(S.shape[0])
        w[idx, idx + diag_off] -= 2.0
        return (coef * (w @ b16.double())).float()


def _fused_case(rank, world):
    from mclstexp_amd import dist as mdist
    b_loc, P, T = 6, 32, 0.7
    g = torch.Generator().manual_seed(4321)
    es_all = torch.nn.functional.layer_norm(torch.randn(world * b_loc, P, generator=g), (P,)).bfloat16().float()
    ei_all = torch.nn.functional.layer_norm(torch.randn(world * b_loc, P, generator=g), (P,)).bfloat16().float()
    sl = slice(rank * b_loc, (rank + 1) * b_loc)
    loss, d_es, d_ei, _ = mdist.dist_infonce_fused_fwd_bwd(es_all[sl].clone(), ei_all[sl].clone(), T, td.group.WORLD,
                                                           prims=OracleFusedPrims)
    return loss.item(), d_es, d_ei


@pytest.mark.parametrize("world", [2, 4])
def test_dist_fused_infonce_equals_single_process(world):
    """The fused data-parallel decomposition (bf16 all-gather, two strip orientations per rank, LSE exchange)
    reproduces the single-process loss and gradients on bf16-representable embeddings."""
    outs = _spawn(_fused_case, world)
    b_loc, P, T = 6, 32, 0.7
    g = torch.Generator().manual_seed(4321)
    es = torch.nn.functional.layer_norm(torch.randn(world * b_loc, P, generator=g), (P,)).bfloat16().float()
    ei = torch.nn.functional.layer_norm(torch.randn(world * b_loc, P, generator=g), (P,)).bfloat16().float()
    es.requires_grad_(True); ei.requires_grad_(True)
    loss = ref_cpu.symmetric_infonce(ref_cpu.logits(es, ei, T))
    loss.backward()
    for r, (l, d_es, d_ei) in enumerate(outs):
        sl = slice(r * b_loc, (r + 1) * b_loc)
        assert abs(l - loss.item()) < 1e-5
        assert torch.allclose(d_es, es.grad[sl], atol=2e-6), (d_es - es.grad[sl]).abs().max()
        assert torch.allclose(d_ei, ei.grad[sl], atol=2e-6)


def _rows_case(rank, world):
    from mclstexp_amd import dist as mdist
    g = torch.Generator().manual_seed(7 + rank)
    dout = torch.randn(5, 11, generator=g)
    ix = torch.randint(0, 9, (5,), generator=g, dtype=torch.int32)
    iy = torch.randint(0, 9, (5,), generator=g, dtype=torch.int32)
    G, X, Y = mdist.gather_rows(dout, ix, iy, td.group.WORLD)
    return dout, ix, iy, G, X, Y


def test_gather_rows_is_rank_ordered_concat():
    outs = _spawn(_rows_case, 2)
    dcat = torch.cat([o[0] for o in outs]); xcat = torch.cat([o[1] for o in outs]); ycat = torch.cat([o[2] for o in outs])
    for o in outs:
        assert torch.equal(o[3], dcat) and torch.equal(o[4], xcat) and torch.equal(o[5], ycat)


RAGGED = {2: [6, 3], 4: [5, 2, 7, 4], 8: [4, 4, 1, 4, 3, 4, 4, 2]}


def _ragged_case(rank, world):
    """Ragged shards (last batch of an epoch, train.py:49 has no drop_last): sizes agreed on the host."""
    from mclstexp_amd import dist as mdist
    sizes = mdist.SizeExchange(td.group.WORLD)(RAGGED[world][rank])
    assert sizes == RAGGED[world]
    mdist.set_step_sizes(sizes)
    P, T = 32, 0.7
    n = sum(sizes)
    g = torch.Generator().manual_seed(77)
    es_all = torch.nn.functional.layer_norm(torch.randn(n, P, generator=g), (P,)).bfloat16().float()
    ei_all = torch.nn.functional.layer_norm(torch.randn(n, P, generator=g), (P,)).bfloat16().float()
    o = sum(sizes[:rank])
    sl = slice(o, o + sizes[rank])
    le, d_es, d_ei, s_rows = mdist.dist_infonce_fwd_bwd(es_all[sl].clone(), ei_all[sl].clone(), T, td.group.WORLD,
                                                        prims=OraclePrims)
    lf, f_es, f_ei, _ = mdist.dist_infonce_fused_fwd_bwd(es_all[sl].clone(), ei_all[sl].clone(), T, td.group.WORLD,
                                                         prims=OracleFusedPrims)
    g2 = torch.Generator().manual_seed(100 + rank)
    dout = torch.randn(sizes[rank], 11, generator=g2)
    ix = torch.randint(0, 9, (sizes[rank],), generator=g2, dtype=torch.int32)
    iy = torch.randint(0, 9, (sizes[rank],), generator=g2, dtype=torch.int32)
    G, X, Y = mdist.gather_rows(dout, ix, iy, td.group.WORLD)
    mdist.set_step_sizes(None)
    return le.item(), d_es, d_ei, lf.item(), f_es, f_ei, (dout, ix, iy, G, X, Y)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ragged_shards_equal_single_process(world):
    outs = _spawn(_ragged_case, world)
    sizes = RAGGED[world]
    P, T = 32, 0.7
    n = sum(sizes)
    g = torch.Generator().manual_seed(77)
    es = torch.nn.functional.layer_norm(torch.randn(n, P, generator=g), (P,)).bfloat16().float().requires_grad_(True)
    ei = torch.nn.functional.layer_norm(torch.randn(n, P, generator=g), (P,)).bfloat16().float().requires_grad_(True)
    loss = ref_cpu.symmetric_infonce(ref_cpu.logits(es, ei, T))
    loss.backward()
    dcat = torch.cat([o[6][0] for o in outs]); xcat = torch.cat([o[6][1] for o in outs]); ycat = torch.cat([o[6][2] for o in outs])
    for r, (le, d_es, d_ei, lf, f_es, f_ei, rows) in enumerate(outs):
        o = sum(sizes[:r])
        sl = slice(o, o + sizes[r])
        assert abs(le - loss.item()) < 1e-5 and abs(lf - loss.item()) < 1e-5
        assert torch.allclose(d_es, es.grad[sl], atol=2e-6) and torch.allclose(d_ei, ei.grad[sl], atol=2e-6)
        assert torch.allclose(f_es, es.grad[sl], atol=2e-6) and torch.allclose(f_ei, ei.grad[sl], atol=2e-6)
        assert torch.equal(rows[3], dcat) and torch.equal(rows[4], xcat) and torch.equal(rows[5], ycat)


def _bad_sizes_case(rank, world):
    from mclstexp_amd import dist as mdist
    try:
        mdist._all_gather_rows(torch.zeros(3, 2), td.group.WORLD, sizes=[4, 4])
    except RuntimeError as e:
        return str(e)
    return ""


def test_ragged_all_gather_rejects_wrong_sizes():
    for msg in _spawn(_bad_sizes_case, 2):
        assert "ragged all-gather" in msg


class _FakeFlatOpt:
    """Stands in for FusedAdam's flat-bucket interface (its step() needs the GPU)."""

    def __init__(self, rank):
        self.flat = torch.full((10,), float(rank + 1))
        self.a = torch.nn.Parameter(torch.zeros(3)); self.a.grad = self.flat[:3]
        self.b = torch.nn.Parameter(torch.zeros(2)); self.b.grad = torch.full((2,), 10.0 * (rank + 1))
        self.c = torch.nn.Parameter(torch.zeros(2))                       # no grad: skipped
        self.param_groups = [{"params": [self.a, self.b, self.c]}]

    def ensure_flat(self):
        pass

    def flat_grads(self):
        return [self.flat]

    def flat_param_ids(self):
        return {id(self.a)}


def _reduce_case(rank, world):
    from mclstexp_amd import dist as mdist
    opt = _FakeFlatOpt(rank)
    mdist.GradReducer(td.group.WORLD).reduce(opt)
    return opt.flat.clone(), opt.b.grad.clone()


def test_grad_reducer_sums_flat_bucket_and_stragglers():
    for flat, b in _spawn(_reduce_case, 2):
        assert torch.equal(flat, torch.full((10,), 3.0))       # 1 + 2: SUM, not mean
        assert torch.equal(b, torch.full((2,), 30.0))


def _bucket_case(rank, world):
    from mclstexp_amd import dist as mdist
    g = torch.Generator().manual_seed(100 + rank)
    base = torch.randn(100003, generator=g)

    class Opt(_FakeFlatOpt):
        def __init__(self, flat):
            self.flat = flat
            self.param_groups = [{"params": []}]

        def flat_param_ids(self):
            return set()

    outs = {}
    for name, kw in (("mono", dict(buckets=1)), ("b4", dict(buckets=4)), ("b7", dict(buckets=7)),
                     ("bf16", dict(buckets=3, wire="bf16"))):
        opt = Opt(base.clone())
        handles = mdist.GradReducer(td.group.WORLD, **kw).reduce(opt, async_flat=True)
        for h in handles:
            h.wait()
        outs[name] = opt.flat.clone()
    return outs, base


def test_grad_reducer_buckets_equal_monolithic_and_bf16_wire():
    """Bucketed all-reduce == monolithic bit for bit (world 2: each element is one commutative fp32 addition whatever the
    chunking); bf16 wire format = the sum of the bf16-rounded summands, rounded to bf16."""
    res = _spawn(_bucket_case, 2)
    (o0, b0), (o1, b1) = res
    exact = b0 + b1
    for o in (o0, o1):
        assert torch.equal(o["mono"], exact)
        assert torch.equal(o["b4"], o["mono"]) and torch.equal(o["b7"], o["mono"])
        ref16 = (b0.to(torch.bfloat16) + b1.to(torch.bfloat16)).float()
        assert torch.equal(o["bf16"], ref16)
        assert (o["bf16"] - exact).abs().max() <= 2.0 ** -7 * exact.abs().max()
    assert torch.equal(o0["mono"], o1["mono"]) and torch.equal(o0["bf16"], o1["bf16"])          # replicas identical


def _range_case(rank, world):
    from mclstexp_amd import dist as mdist
    g = torch.Generator().manual_seed(300 + rank)
    base = torch.randn(50007, generator=g)
    red = mdist.GradReducer(td.group.WORLD)
    mono = base.clone()
    for h in red.reduce_range(mono, 0, mono.numel(), async_flat=True):
        h.wait()
    parts = base.clone()
    handles = []
    bounds = [50007, 41000, 20004, 8, 0]                  # tail first, like the backward segments of engine.TrainStep
    for hi, lo in zip(bounds, bounds[1:]):
        handles += red.reduce_range(parts, lo, hi, async_flat=True)
    handles += red.reduce_range(parts, 5, 5)              # an empty range is a no-op
    for h in handles:
        h.wait()
    return mono, parts, base


def test_grad_reducer_ranges_equal_monolithic():
    """One all-reduce per gradient range (the segmented data-parallel backward) == one all-reduce of the whole bucket, bit
    for bit, and the replicas agree."""
    (m0, p0, b0), (m1, p1, b1) = _spawn(_range_case, 2)
    assert torch.equal(m0, b0 + b1) and torch.equal(p0, m0)
    assert torch.equal(m1, m0) and torch.equal(p1, p0)


def test_init_from_env_single_process(monkeypatch):
    from mclstexp_amd import dist as mdist
    monkeypatch.setenv("WORLD_SIZE", "1")
    pg, rank, world = mdist.init_from_env()
    assert pg is None and rank == 0 and world == 1

"""Inference-time retrieval on the MI355X against the reference's own outputs
(tests/golden/retrieval_*.npz) and the numpy oracle (oracle/ref_retrieval.py).  pytest -m gpu."""
import numpy as np
import pytest
import torch

from helpers import RETRIEVAL_CASES, assert_close, assert_close_scaled, check_topk, load_retrieval_golden
from mclstexp_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def rt():
    from mclstexp_amd import _lib, retrieval
    _lib.lib()  # must load: no fallback
    return retrieval


def _cpu_topk(sim: np.ndarray, k: int):
    """Exact reference ordering on a GIVEN fp32 matrix: value descending, index ascending among equals."""
    order = np.lexsort((np.arange(sim.shape[1])[None, :].repeat(sim.shape[0], 0), -sim.astype(np.float64)), axis=1)
    idx = order[:, :k]
    return np.take_along_axis(sim, idx, axis=1), idx


# ------------------------------------------------------------------ kernels
@pytest.mark.parametrize("rows,dim", [(1, 256), (33, 256), (257, 100), (5, 7)])
def test_l2_normalize_rows(rt, rows, dim):
    g = torch.Generator().manual_seed(rows * 1000 + dim)
    x = torch.randn(rows, dim, generator=g) * 3
    x[0] = 0.0                                        # zero row: F.normalize's eps clamp -> zeros, not NaN
    y = rt.l2_normalize(x.to(DEV)).cpu()
    ref = torch.nn.functional.normalize(x.double(), p=2, dim=-1)
    assert_close(y, ref, 2e-7, what="l2_normalize")
    assert torch.equal(y[0], torch.zeros(dim))


TOPK_SHAPES = [(1, 1, 1), (3, 7, 7), (4, 100, 1), (9, 1000, 200), (5, 3001, 600), (2, 100003, 2048), (6, 513, 512)]


@pytest.mark.parametrize("rows,n,k", TOPK_SHAPES)
def test_topk_rows_exact_on_given_matrix(rt, rows, n, k):
    """Selection on a given fp32 matrix has no rounding: values AND indices must equal the CPU ordering bit for bit."""
    g = torch.Generator().manual_seed(n + k)
    sim = (torch.rand(rows, n, generator=g) * 2 - 1)
    if rows > 1:
        sim[1] = sim[1] * 1e-3 + 0.75               # narrow value range (what cosine rows look like)
    sim_np = sim.numpy()
    v, i = rt.topk_rows(sim.to(DEV), k)
    rv, ri = _cpu_topk(sim_np, k)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(v.cpu().numpy(), rv)


def test_topk_rows_ties_specials_and_strides(rt):
    n, k = 5000, 300
    g = torch.Generator().manual_seed(7)
    base = torch.rand(8, n, generator=g)
    base[0] = 0.5                                                       # all equal: lowest k indices
    base[1] = torch.floor(base[1] * 8) / 8                              # 8 distinct values: ties across the k-th
    base[2, ::3] = -base[2, ::3]                                        # mixed signs
    base[3, 10] = float("inf"); base[3, 20] = float("-inf"); base[3, 30] = 0.0; base[3, 31] = -0.0
    base[4] = -base[4] - 5.0                                            # all negative
    base[5] = torch.floor(base[5] * 2)                                  # two values
    base[6, :] = 1.0; base[6, 4000:] = 2.0                              # winners at the end + ties at the front
    base[7] = base[7] * 1e-30                                           # tiny magnitudes (exponent-dominated keys)
    wide = torch.zeros(8, n + 13)
    wide[:, :n] = base
    wide[:, n:] = 9.0                                                   # beyond n: must never be read as data
    v, i = rt.topk_rows(wide.to(DEV)[:, :n], k)                         # leading dimension n + 13
    rv, ri = _cpu_topk(base.numpy(), k)
    # -0.0 == 0.0 for torch.topk but not for the integer keys: compare that row by value only
    vi, ii = v.cpu().numpy(), i.cpu().numpy()
    for r in range(8):
        assert np.array_equal(vi[r], rv[r]), f"row {r} values"
        if r != 3:
            assert np.array_equal(ii[r], ri[r]), f"row {r} indices"
    assert ii[0].tolist() == list(range(k))
    assert ii[3, 0] == 10 and 20 not in ii[3].tolist()


def test_topk_rows_argument_errors(rt):
    sim = torch.zeros(2, 10, device=DEV)
    with pytest.raises(RuntimeError):
        rt.topk_rows(sim, 11)
    with pytest.raises(RuntimeError):
        rt.topk_rows(torch.zeros(2, 5000, device=DEV), 4000)           # > mcl_topk_rows_max_k()
    with pytest.raises(RuntimeError):
        rt.topk_rows(torch.zeros(2, 10), 3)                              # CPU tensor: no fallback


# ------------------------------------------------------------------ against the reference's outputs
@pytest.mark.parametrize("name", sorted(RETRIEVAL_CASES))
def test_find_matches_against_reference_fixture(rt, name):
    from oracle import ref_retrieval
    z, case, meta = load_retrieval_golden(name)
    sim64 = ref_retrieval.similarity_f64(case["spot_key"], case["image_query"])
    values, indices = rt.find_matches(case["spot_key"], case["image_query"], top_k=meta["top_k"], return_values=True)
    assert indices.dtype == np.int64 and indices.shape == z["indices"].shape
    # tolerance-aware exactness (tests/helpers.check_topk): 2e-6 on cosine values, fp32 sums of 256 products
    exact_rows = check_topk(indices, sim64, tol=2e-6, values=values, what=name)
    assert exact_rows >= int(0.9 * meta["Q"]), f"only {exact_rows}/{meta['Q']} rows match the fp64 sets exactly"
    # and directly against the reference's indices: identical except where fp32 near-ties reorder
    assert (indices == z["indices"]).mean() > 0.97
    if "values" in z.files:
        assert_close(values, z["values"], 1e-6, what="top-k similarities (evel_cscc returns them)")
    assert np.array_equal(rt.find_matches(case["spot_key"], case["image_query"], top_k=meta["top_k"]), indices)


@pytest.mark.parametrize("name", sorted(RETRIEVAL_CASES))
def test_weighted_average_against_reference_fixture(rt, name):
    """The weighting loop on the reference's own indices: fp32 inputs, fp64 accumulation on the device; the
    reference's numpy accumulates fp32 products pairwise, so 5e-6 of the row's largest value is the bar."""
    z, case, meta = load_retrieval_golden(name)
    emb, expr = rt.weighted_average_device(case["spot_key"], case["expression_key"], case["image_query"],
                                           z["indices"], ord=meta["ord"])
    assert_close_scaled(emb.cpu().numpy(), z["emb_pred"], 5e-6, what="matched_spot_embeddings_pred")
    assert_close_scaled(expr.cpu().numpy(), z["expr_pred"], 5e-6, what="matched_spot_expression_pred")
    # embeddings only (expression_key=None)
    emb2, none = rt.weighted_average_device(case["spot_key"], None, case["image_query"], z["indices"], ord=meta["ord"])
    assert none is None and torch.equal(emb2, emb)


@pytest.mark.parametrize("name", ["her2st", "cscc"])
def test_predict_expression_end_to_end(rt, name):
    z, case, meta = load_retrieval_golden(name)
    out = rt.predict_expression(case["spot_key"], case["expression_key"], case["image_query"], top_k=meta["top_k"],
                                ord=meta["ord"])
    assert out["matched_spot_expression_pred"].dtype == np.float64
    same = np.array([set(a.tolist()) == set(b.tolist()) for a, b in zip(out["indices"], z["indices"])])
    assert same.mean() >= 0.9
    assert_close_scaled(out["matched_spot_expression_pred"][same], z["expr_pred"][same], 5e-6, what="expr_pred")
    assert_close_scaled(out["matched_spot_embeddings_pred"][same], z["emb_pred"][same], 5e-6, what="emb_pred")
    # a near-tie swap at rank k exchanges one of k neighbours with an almost equally distant one
    assert_close_scaled(out["matched_spot_expression_pred"], z["expr_pred"], 2e-2, what="expr_pred (all rows)")


def test_exact_match_gives_nan_row_like_numpy(rt):
    from oracle import ref_retrieval
    case = synth.make_retrieval_case(400, 6, 256, 31, seed=9)
    q = case["image_query"].copy()
    q[2] = case["spot_key"][17]                                         # distance 0 -> 1/0 -> inf/inf
    _, idx = ref_retrieval.find_matches(case["spot_key"], q, top_k=20)
    with np.errstate(all="ignore"):
        ref_emb, ref_expr = ref_retrieval.weighted_prediction(case["spot_key"], case["expression_key"], q, idx, ord=2)
    emb, expr = rt.weighted_average_device(case["spot_key"], case["expression_key"], q, idx, ord=2)
    assert np.isnan(ref_expr[2]).all() and torch.isnan(expr[2]).all() and torch.isnan(emb[2]).all()
    keep = [0, 1, 3, 4, 5]
    assert_close_scaled(expr.cpu().numpy()[keep], ref_expr[keep], 5e-6, what="other rows")


def test_duplicate_keys_tie_order(rt):
    """Exact duplicates among the keys (ties in the similarity): the copies are interchangeable for the reference
    (torch.topk's tie order is unspecified); here they come out in ascending index order, deterministically."""
    case = synth.make_retrieval_case(1200, 9, 256, 11, seed=5, duplicates=40)
    q = case["image_query"].copy()
    q[0] = case["spot_key"][0] * 1.5                                    # cosine 1 with key 0 and its 40 copies
    v, i = rt.find_matches(case["spot_key"], q, top_k=30, return_values=True)
    dup = [0] + list(range(1160, 1200))
    assert i[0].tolist() == dup[:30]
    v2, i2 = rt.find_matches(case["spot_key"], q, top_k=30, return_values=True)
    assert np.array_equal(i, i2) and np.array_equal(v, v2)


# ------------------------------------------------------------------ size-independent properties at full size
def test_topk_properties_at_scale(rt, monkeypatch):
    monkeypatch.setattr(rt, "FUSED_MIN_SIMS", 1)                     # through the filtered path (find_matches_device's own
    n, q, k, p = 65536, 512, 600, 256                                   # threshold is 1e8 similarities)
    g = torch.Generator(device=DEV).manual_seed(3)
    keys = torch.randn(n, p, device=DEV, generator=g)
    query = torch.randn(q, p, device=DEV, generator=g) + 0.5 * keys[:q]
    v, i = rt.find_matches_device(keys, query, k)
    sim = rt.cosine_similarity_matrix(rt.l2_normalize(query), rt.l2_normalize(keys))
    assert (v[:, 1:] <= v[:, :-1]).all()                                           # best first
    assert torch.equal(torch.gather(sim, 1, i), v)                                 # values are the matrix entries
    assert ((sim > v[:, -1:]).sum(1) <= k - 1).all() and ((sim >= v[:, -1:]).sum(1) >= k).all()   # exact k-th
    assert (torch.sort(i, 1).values[:, 1:] != torch.sort(i, 1).values[:, :-1]).all()              # no repeats
    assert (i[:, 0] == torch.arange(q, device=DEV)).all()                          # the planted neighbour wins
    # permuting the keys permutes the answer
    perm = torch.randperm(n, device=DEV, generator=g)
    v2, i2 = rt.find_matches_device(keys[perm], query, k)
    assert torch.equal(v2, v)
    assert torch.equal(torch.sort(perm[i2], 1).values, torch.sort(i, 1).values)
    # weighted average: weights are a convex combination -> prediction inside the neighbours' range; constant
    # expression is reproduced
    expr = torch.full((n, 40), 2.5, device=DEV)
    emb, ex = rt.weighted_average_device(keys, expr, query, i, ord=2)
    assert_close(ex.cpu().numpy(), np.full((q, 40), 2.5), 1e-6, what="constant expression")
    nb = keys[i[:4]]                                                               # (4, k, p)
    assert (emb[:4] <= nb.max(1).values + 1e-5).all() and (emb[:4] >= nb.min(1).values - 1e-5).all()


# ------------------------------------------------------------------ fused similarity + top-k (no (Q, N) matrix in HBM)
def _materialised(rt, query, keys, k):
    return rt.topk_rows(rt.cosine_similarity_matrix(query, keys), k)


@pytest.mark.parametrize("n,q,k", [(20000, 300, 200), (65536, 257, 600), (8192, 64, 1), (100000, 128, 50)])
def test_find_matches_filtered_equals_materialised(rt, n, q, k):
    """mcl_gemm's threshold-filter epilogue + mcl_topk_rows_indexed against the materialised similarity + mcl_topk_rows:
    the same values bit for bit and the same indices, no row recomputed on well-mixed data."""
    g = torch.Generator(device=DEV).manual_seed(n + k)
    keys = rt.l2_normalize(torch.randn(n, 256, device=DEV, generator=g))
    query = rt.l2_normalize(torch.randn(q, 256, device=DEV, generator=g) + 0.3 * keys[torch.randint(0, n, (q,), device=DEV, generator=g)])
    v1, i1, redo = rt.find_matches_filtered(query, keys, k)
    v0, i0 = _materialised(rt, query, keys, k)
    assert torch.equal(v1, v0) and torch.equal(i1, i0)
    assert redo == 0
    v2, i2, _ = rt.find_matches_filtered(query, keys, k)
    assert torch.equal(v2, v1) and torch.equal(i2, i1)                      # run to run (the candidate order is not fixed)


def test_find_matches_filtered_beyond_262k_keys_and_in_query_chunks(rt, monkeypatch):
    """with a fixed 4096-key sample the expected list length 1.5 k + 32 N / 4096 passed the capacity from
    N ~ 262k on and EVERY row fell back to an unchunked (Q, N) recomputation.  The sample now grows with N; and queries (and any
    recomputation) are processed in SIM_WORKSPACE_BYTES-sized chunks."""
    n, q, k = 600_000, 96, 20
    g = torch.Generator(device=DEV).manual_seed(5)
    keys = rt.l2_normalize(torch.randn(n, 256, device=DEV, generator=g))
    query = rt.l2_normalize(torch.randn(q, 256, device=DEV, generator=g))
    v1, i1, redo = rt.find_matches_filtered(query, keys, k)
    assert redo == 0
    v0, i0 = _materialised(rt, query, keys, k)
    assert torch.equal(v1, v0) and torch.equal(i1, i0)
    # a workspace bound that forces several query chunks AND row-by-row recomputation of the rows the threshold missed
    ns, _, _, cap = rt._filter_plan(n, k)
    monkeypatch.setattr(rt, "SIM_WORKSPACE_BYTES", 7 * (4 * ns + 8 * cap))
    query[5] = keys[::n // ns][:ns][:300].sum(0)                            # close to many SAMPLED keys: threshold too high
    query = rt.l2_normalize(query)
    v2, i2, redo2 = rt.find_matches_filtered(query, keys, k)
    v0, i0 = _materialised(rt, query, keys, k)
    assert torch.equal(v2, v0) and torch.equal(i2, i0)


def test_find_matches_filtered_recomputes_rows_the_threshold_missed(rt):
    """Exactness does not depend on the sample: (a) the sampled keys are the query's BEST ones (threshold too high: fewer than k
    candidates), (b) a cluster of near-duplicates overflows a list, (c) exact duplicates tie at the k-th value (the cut must
    follow the key index) -- those rows are recomputed on the materialised path and every result equals it."""
    n, q, k = 16384, 48, 100
    g = torch.Generator(device=DEV).manual_seed(7)
    keys = torch.randn(n, 256, device=DEV, generator=g)
    query = torch.randn(q, 256, device=DEV, generator=g)
    step = n // min(n, rt.FUSED_SAMPLE)
    keys[::step][: rt.FUSED_SAMPLE] += 3.0 * query[0]                      # (a) every sampled key is close to query 0
    keys[5000:5000 + 9000:1] = keys[5000:5000 + 9000] * 0.05 + query[1]    # (b) 9000 near-copies of query 1: list overflow
    keys[200:260] = keys[200]                                              # (c) 60 exact duplicates ...
    query[2] = keys[200] + 0.01 * torch.randn(256, device=DEV, generator=g)   # ... that are query 2's best matches, cut at k = 100? no:
    keys[300:420] = keys[300]                                              #     120 duplicates: the k-th value IS inside the tie
    query[3] = keys[300]
    kn, qn = rt.l2_normalize(keys), rt.l2_normalize(query)
    v1, i1, redo = rt.find_matches_filtered(qn, kn, k)
    v0, i0 = _materialised(rt, qn, kn, k)
    assert redo >= 2
    assert torch.equal(v1, v0) and torch.equal(i1, i0)
    assert i1[3].tolist() == list(range(300, 400))                          # ties cut by key index, as the materialised path does


# ------------------------------------------------------------------ embedding extraction (evel_her2st.py:41-69)
def test_get_embeddings_matches_reference_fixture_and_oracle(rt):
    """Eval-mode, no_grad sub-module calls in the reference's order.  First batch = the golden case (embeddings
    produced by the reference's own classes), second batch ragged (B = 5) against the oracle."""
    from helpers import load_golden, oracle_forward
    from mclstexp_amd.model import mclSTExp_Attention
    z, meta = load_golden("b8_g785")
    G, D, L = meta["G"], meta["D"], meta["layers"]
    params = synth.make_params(G, D, 256, 8, 64, L, seed=0)
    m = mclSTExp_Attention("identity", meta["T"], D, G, 256, 8, 64, L)
    m.load_state_dict(params, strict=True)
    m.to(DEV)
    b0 = synth.make_batch(meta["B"], G, image_dim=D, seed=0)
    b1 = synth.make_batch(5, G, image_dim=D, seed=3)
    img, spot = rt.get_embeddings(m, [b0, b1])
    assert not m.training and not img.requires_grad
    assert img.shape == (meta["B"] + 5, 256) and spot.shape == (meta["B"] + 5, 256)
    assert_close(img[:meta["B"]].cpu(), z["image_embeddings"], 2e-5, what="image_embeddings (reference fixture)")
    assert_close(spot[:meta["B"]].cpu(), z["spot_embeddings"], 2e-5, what="spot_embeddings (reference fixture)")
    with torch.no_grad():
        ref = oracle_forward(params, b1, meta)
    assert_close(img[meta["B"]:].cpu(), ref["image_embeddings"], 2e-5, what="image_embeddings (oracle, ragged)")
    assert_close(spot[meta["B"]:].cpu(), ref["spot_embeddings"], 2e-5, what="spot_embeddings (oracle, ragged)")
    # and straight into the retrieval: every spot retrieves itself first among its own embeddings
    idx = rt.find_matches(spot, spot, top_k=3)
    assert idx[:, 0].tolist() == list(range(meta["B"] + 5))

"""Inference-time retrieval on the MI355X: the reference's ``get_embeddings`` / ``find_matches`` / weighting loop
 with the same names and argument meaning, executed by HIP kernels through the C ABI.

Reference (paths relative to /root/reference/):
  get_embeddings   evel_her2st.py:30-69   (eval-mode sub-module calls, no_grad)
  find_matches     evel_her2st.py:74-84, evel_visium.py:94-104 (indices), evel_cscc.py:74-84 (values, indices)
  weighting loop   evel_her2st.py:174-187 (top 200, L1), evel_cscc.py:197-215 (top 600, L2),
                   evel_visium.py:193-205 (top 200, L2)

The reference works on numpy arrays on the host (``torch.tensor(...)`` on CPU, a Python loop over queries);
here the arrays are moved to the GPU once and stay there: L2 normalisation, the (Q, N) cosine similarity as an fp32
MFMA GEMM, an exact per-row radix-select top-k and the weighted neighbour average are four launches per query
chunk; for large problems (>= 1e8 similarities) the similarity matrix is never written (``find_matches_filtered``: a per-query threshold from a key
sample, the GEMM's filter epilogue, top-k of the candidate lists -- exact, with a materialised recomputation of any row the
threshold missed).  No CPU fallback: without a GPU / the HIP library these functions raise ``RuntimeError``.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Tuple, Union

import numpy as np
import torch

from . import _lib, ops
from ._lib import check

Tensor = torch.Tensor
ArrayLike = Union[np.ndarray, Tensor]

# upper bound of the (Q_chunk, N) fp32 similarity workspace
SIM_WORKSPACE_BYTES = 1 << 30


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("mclstexp_amd.retrieval: no GPU available (HIP kernels, no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(a: ArrayLike, name: str) -> Tensor:
    t = torch.as_tensor(a) if not isinstance(a, Tensor) else a
    if t.dim() != 2:
        raise RuntimeError(f"{name}: expected a 2-D array, got shape {tuple(t.shape)}")
    if not t.is_cuda:
        t = t.to(_device(), non_blocking=False)
    t = t.to(torch.float32)
    return ops._rowmajor(t, name)


def l2_normalize(x: Tensor) -> Tensor:
    """F.normalize(x, p=2, dim=-1) (evel_her2st.py:78-79)."""
    x = ops._rowmajor(x, "x")
    y = torch.empty_like(x, memory_format=torch.contiguous_format)
    check(_lib.lib().mcl_l2_normalize_rows(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), x.shape[0],
                                           x.shape[1], ops._stream()), "mcl_l2_normalize_rows")
    return y


def cosine_similarity_matrix(query_n: Tensor, keys_n: Tensor) -> Tensor:
    """``query_n @ keys_n.T`` (evel_her2st.py:80) on already-normalised rows: fp32 MFMA (exact fp32 products), whatever
    compute mode the training path is set to."""
    m, p = query_n.shape
    n = keys_n.shape[0]
    sim = torch.empty((m, n), device=query_n.device, dtype=torch.float32)
    ops.gemm_raw(m, n, p, 1, query_n, query_n.stride(0), 1, 0, keys_n, 1, keys_n.stride(0), 0, sim, n, 0,
                 compute=_lib.COMPUTE_F32)
    return sim


def topk_rows(sim: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """torch.topk(sim, k) along the last dim of a 2-D fp32 matrix: (values, int64 indices), best first."""
    sim = ops._rowmajor(sim, "sim")
    rows, n = sim.shape
    if not 0 < k <= n:
        raise RuntimeError(f"top_k={k} out of range for {n} keys")
    values = torch.empty((rows, k), device=sim.device, dtype=torch.float32)
    indices = torch.empty((rows, k), device=sim.device, dtype=torch.int64)
    check(_lib.lib().mcl_topk_rows(sim.data_ptr(), sim.stride(0), rows, n, k, values.data_ptr(), indices.data_ptr(),
                                   ops._stream()), "mcl_topk_rows")
    return values, indices


# The fused similarity + top-k path (find_matches_filtered) is taken when the similarity matrix would be large (its extra
# launches and the host look at the counters cost ~0.2 ms: at the reference's fold sizes, 1e7 similarities, the materialised
# path is faster -- tools/bench_retrieval.py) and k is a small part of the keys
FUSED_MIN_KEYS = 8192
FUSED_MIN_SIMS = 100_000_000
FUSED_SAMPLE = 4096          # smallest key sample
FUSED_MAX_STRIDE = 64        # ... and the sample grows with N so that it is never sparser than every 64th key
FUSED_RANK_SLACK = 32        # sample ranks of safety below the expected rank of the k-th best


def _filter_plan(n: int, top_k: int) -> Tuple[int, int, int, int]:
    """(sample size ns, sample stride, threshold rank r in the sample, candidate capacity per query) of the filtered path.
    The threshold sits at sample rank r = 1.5 k ns / N + slack, so a row's list is expected to hold r N / ns = 1.5 k + slack
    N / ns candidates -- the slack term grows with the sample's sparsity (with a fixed 4096-key sample it passed any
    fixed capacity from N ~ 262k on and EVERY row was recomputed).  The sample therefore grows with N (stride <= 64) and the
    capacity is twice the expectation + 1024: the count's relative spread is ~ 1 / sqrt(r) <= 18 %."""
    ns = min(n, max(FUSED_SAMPLE, -(-n // FUSED_MAX_STRIDE)))
    step = n // ns
    r = min(ns, int(1.5 * top_k * ns / n) + FUSED_RANK_SLACK)
    expect = r * n / ns
    cap = int(2 * expect) + 1024
    return ns, step, r, cap


def find_matches_filtered(query: Tensor, keys: Tensor, top_k: int) -> Tuple[Tensor, Tensor, int]:
    """Cosine top-k WITHOUT the (Q, N) similarity matrix in HBM (evel_her2st.py:74-84), exact:
      1. a strided sample of the keys (>= FUSED_SAMPLE, at least every 64th) gives every query a threshold -- the value at rank
         ~1.5 k N_s / N of its sample similarities (small GEMM + mcl_topk_rows);
      2. the full similarity GEMM runs with mcl_gemm's FILTER epilogue: products >= the row's threshold are appended to the row's
         candidate list, nothing else is stored;
      3. mcl_topk_rows_indexed selects the top-k of every list (original key indices, equal values ordered by index).
    A row whose list holds fewer than k or more than its capacity, or whose k-th value is an exact tie, is recomputed on the
    materialised path -- the result is the same as ``topk_rows(cosine_similarity_matrix(...))`` for every input.  Queries are
    processed in chunks so that neither the sample similarities + candidate lists nor a recomputation ever exceed
    SIM_WORKSPACE_BYTES.  Returns (values, indices, number of recomputed rows)."""
    q, p = query.shape
    n = keys.shape[0]
    dev = query.device
    L = _lib.lib()
    ns, step, r, cap = _filter_plan(n, top_k)
    sample = keys[::step][:ns]                                   # strided view: rows stay unit-stride
    values = torch.empty((q, top_k), device=dev, dtype=torch.float32)
    indices = torch.empty((q, top_k), device=dev, dtype=torch.int64)
    chunk = max(1, min(q, SIM_WORKSPACE_BYTES // (4 * ns + 8 * cap)))
    redo_chunk = max(1, SIM_WORKSPACE_BYTES // (4 * n))
    redone = 0
    for q0 in range(0, q, chunk):
        q1 = min(q, q0 + chunk)
        qc = q1 - q0
        qry = query[q0:q1]
        thr = topk_rows(cosine_similarity_matrix(qry, sample), r)[0][:, r - 1].contiguous()
        cnt = torch.zeros((qc,), device=dev, dtype=torch.int32)
        cval = torch.full((qc, cap), float("-inf"), device=dev, dtype=torch.float32)
        cidx = torch.zeros((qc, cap), device=dev, dtype=torch.int32)
        ops.gemm_raw(qc, n, p, 1, qry, qry.stride(0), 1, 0, keys, 1, keys.stride(0), 0, None, n, 0,
                     compute=_lib.COMPUTE_F32, filt=(thr, cnt, cval, cidx))
        v, i = values[q0:q1], indices[q0:q1]
        tie = torch.zeros((qc,), device=dev, dtype=torch.int32)
        check(L.mcl_topk_rows_indexed(cval.data_ptr(), cap, cidx.data_ptr(), cap, qc, cap, top_k, v.data_ptr(), i.data_ptr(),
                                      tie.data_ptr(), ops._stream()), "mcl_topk_rows_indexed")
        bad = ((cnt < top_k) | (cnt > cap) | (tie != 0)).nonzero().flatten()  # (host sync: the retrieval returns to the host anyway)
        for b0 in range(0, int(bad.numel()), redo_chunk):
            rows = bad[b0:b0 + redo_chunk]
            vv, ii = topk_rows(cosine_similarity_matrix(qry[rows].contiguous(), keys), top_k)
            v[rows], i[rows] = vv, ii
        redone += int(bad.numel())
    return values, indices, redone


def find_matches_device(spot_embeddings: ArrayLike, query_embeddings: ArrayLike, top_k: int = 1
                        ) -> Tuple[Tensor, Tensor]:
    """(values, indices) as device tensors, shapes (Q, top_k): cosine top-k of every query against all keys."""
    keys = l2_normalize(_to_dev(spot_embeddings, "spot_embeddings"))
    query = l2_normalize(_to_dev(query_embeddings, "query_embeddings"))
    if keys.shape[1] != query.shape[1]:
        raise RuntimeError(f"embedding widths differ: keys {tuple(keys.shape)}, queries {tuple(query.shape)}")
    q, n = query.shape[0], keys.shape[0]
    if n >= FUSED_MIN_KEYS and q * n >= FUSED_MIN_SIMS and 16 * top_k <= n and top_k <= 1024:
        v, i, _ = find_matches_filtered(query, keys, top_k)
        return v, i
    values = torch.empty((q, top_k), device=keys.device, dtype=torch.float32)
    indices = torch.empty((q, top_k), device=keys.device, dtype=torch.int64)
    chunk = max(1, min(q, SIM_WORKSPACE_BYTES // (4 * n)))
    for q0 in range(0, q, chunk):
        q1 = min(q, q0 + chunk)
        sim = cosine_similarity_matrix(query[q0:q1], keys)
        v, i = topk_rows(sim, top_k)
        values[q0:q1], indices[q0:q1] = v, i
    return values, indices


def find_matches(spot_embeddings: ArrayLike, query_embeddings: ArrayLike, top_k: int = 1,
                 return_values: bool = False):
    """Drop-in for the reference's ``find_matches``: numpy int64 indices (Q, top_k), best match first
    (evel_her2st.py:74-84); with ``return_values`` also the similarities, as evel_cscc.py:74-84 returns them."""
    values, indices = find_matches_device(spot_embeddings, query_embeddings, top_k)
    if return_values:
        return values.cpu().numpy(), indices.cpu().numpy()
    return indices.cpu().numpy()


def weighted_average_device(spot_key: ArrayLike, expression_key: Optional[ArrayLike], image_query: ArrayLike,
                            indices: ArrayLike, ord: int = 2) -> Tuple[Tensor, Optional[Tensor]]:
    """The reference's per-query weighting loop for all queries at once (device tensors, fp32):
    ``a = ||spot_key[idx] - query||_ord``, ``w = a**-2 / sum(a**-2)``, np.average of the matched embeddings and
    expression rows.  ``ord=1``: evel_her2st.py:176; ``ord=2``: evel_cscc.py:209, evel_visium.py:197."""
    key = _to_dev(spot_key, "spot_key")
    qry = _to_dev(image_query, "image_query")
    idx = torch.as_tensor(indices)
    if idx.dim() != 2 or idx.shape[0] != qry.shape[0]:
        raise RuntimeError(f"indices must be (Q, k); got {tuple(idx.shape)} for {qry.shape[0]} queries")
    idx = idx.to(device=key.device, dtype=torch.int64).contiguous()
    if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= key.shape[0]):
        raise IndexError("neighbour index out of range")
    q, k = idx.shape
    emb = torch.empty((q, key.shape[1]), device=key.device, dtype=torch.float32)
    expr_t = expr_out = None
    genes = 0
    if expression_key is not None:
        expr_t = _to_dev(expression_key, "expression_key")
        if expr_t.shape[0] != key.shape[0]:
            raise RuntimeError("expression_key and spot_key must have one row per training spot")
        genes = expr_t.shape[1]
        expr_out = torch.empty((q, genes), device=key.device, dtype=torch.float32)
    check(_lib.lib().mcl_knn_weighted_average(
        key.data_ptr(), key.stride(0), ops._p(expr_t), expr_t.stride(0) if expr_t is not None else 0,
        qry.data_ptr(), qry.stride(0), idx.data_ptr(), q, k, key.shape[1], genes, int(ord), emb.data_ptr(),
        ops._p(expr_out), ops._stream()), "mcl_knn_weighted_average")
    return emb, expr_out


def predict_expression(spot_key: ArrayLike, expression_key: ArrayLike, image_query: ArrayLike, top_k: int = 200,
                       ord: int = 2, method: str = "weighted") -> Dict[str, np.ndarray]:
    """The evaluation section of the reference's eval scripts for one fold (evel_her2st.py:158-187): retrieve the
    ``top_k`` training spots per image query and average them.  Returns numpy arrays ``indices`` (Q, top_k),
    ``matched_spot_embeddings_pred`` (Q, P) and ``matched_spot_expression_pred`` (Q, G), float64 like the
    reference's ``np.zeros`` buffers."""
    if method != "weighted":
        raise ValueError("only the reference's active method 'weighted' is implemented")
    key = _to_dev(spot_key, "spot_key")
    qry = _to_dev(image_query, "image_query")
    _, idx = find_matches_device(key, qry, top_k)
    emb, expr = weighted_average_device(key, expression_key, qry, idx, ord)
    return {"indices": idx.cpu().numpy(),
            "matched_spot_embeddings_pred": emb.cpu().numpy().astype(np.float64),
            "matched_spot_expression_pred": expr.cpu().numpy().astype(np.float64)}


@torch.no_grad()
def get_embeddings(model, loader: Iterable[Dict[str, Tensor]]) -> Tuple[Tensor, Tensor]:
    """(image_embeddings, spot_embeddings), each (N_spots, P): the reference's eval-mode embedding extraction
    (evel_her2st.py:41-69) -- sub-modules called one by one, position tables indexed with ``.long()``."""
    model.eval()
    dev = next(model.parameters()).device
    img_out, spot_out = [], []
    for batch in loader:
        image = batch["image"].to(dev)
        image_features = model.encode_image(image) if hasattr(model, "encode_image") else model.image_encoder(image)
        img_out.append(model.image_projection(image_features))
        spot_feature = batch["expression"].to(dev)
        x = batch["position"][:, 0].long().to(dev)
        y = batch["position"][:, 1].long().to(dev)
        spot_feature = spot_feature + model.x_embed(x) + model.y_embed(y)
        spot_embedding = model.spot_encoder(spot_feature.unsqueeze(dim=0))
        spot_out.append(model.spot_projection(spot_embedding).squeeze(dim=0))
    return torch.cat(img_out), torch.cat(spot_out)

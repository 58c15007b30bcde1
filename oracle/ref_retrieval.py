"""CPU restatement (numpy, as the reference computes it) of mclSTExp's inference-time retrieval:
cosine top-k matching of image-query embeddings against the training spots' embeddings and the
inverse-squared-distance weighted average of the matched spots' expression.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  The product path
(``mclstexp_amd.retrieval``) never imports this file and has no CPU fallback.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks these functions against
``tests/golden/retrieval_*.npz``, produced by ``tests/golden/gen_retrieval_goldens.py`` in the build
container by executing the reference's own ``find_matches`` functions and weighting loops (lifted
from the three ``evel_*.py`` scripts with ``ast``; the scripts themselves cannot be imported -- their
module level reads datasets from absolute Windows paths).

Reference citations are relative to /root/reference/.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch
import torch.nn.functional as F


def find_matches(spot_embeddings: np.ndarray, query_embeddings: np.ndarray, top_k: int = 1
                 ) -> Tuple[np.ndarray, np.ndarray]:
    """(values, indices) of the ``top_k`` most cosine-similar keys per query, best first.

    evel_her2st.py:74-84, evel_visium.py:94-104 (indices only) and evel_cscc.py:74-84 (values too):
    L2-normalise both sides (F.normalize, eps 1e-12), ``query @ key.T``, ``torch.topk``."""
    keys = F.normalize(torch.as_tensor(np.asarray(spot_embeddings)), p=2, dim=-1)
    query = F.normalize(torch.as_tensor(np.asarray(query_embeddings)), p=2, dim=-1)
    dot_similarity = query @ keys.T
    values, indices = torch.topk(dot_similarity, k=top_k)
    return values.numpy(), indices.numpy()


def weighted_prediction(spot_key: np.ndarray, expression_key: np.ndarray, image_query: np.ndarray,
                        indices: np.ndarray, ord: int = 2) -> Tuple[np.ndarray, np.ndarray]:
    """(matched_spot_embeddings_pred (Q,P), matched_spot_expression_pred (Q,G)), float64.

    Per query i: a_j = ||spot_key[idx_ij] - image_query[i]||_ord over the UN-normalised embeddings,
    w_j = a_j^-2 / sum_j a_j^-2, prediction = sum_j w_j * row_j.  ``ord=1`` is her2st
    (evel_her2st.py:174-187), ``ord=2`` cscc / visium (evel_cscc.py:199-215, evel_visium.py:194-205)."""
    q, k = indices.shape
    emb = np.zeros((q, spot_key.shape[1]))
    expr = np.zeros((q, expression_key.shape[1]))
    for i in range(q):
        nb = spot_key[indices[i, :], :]
        a = np.linalg.norm(nb - image_query[i, :], axis=1, ord=(1 if ord == 1 else None))
        r = np.reciprocal(a ** 2)
        w = r / np.sum(r)
        emb[i, :] = np.average(nb, axis=0, weights=w)
        expr[i, :] = np.average(expression_key[indices[i, :], :], axis=0, weights=w)
    return emb, expr


def similarity_f64(spot_key: np.ndarray, image_query: np.ndarray) -> np.ndarray:
    """float64 cosine similarities (Q,N): the yardstick that decides which top-k differences are
    genuine and which are fp32 near-ties (used by the parity tests only)."""
    kk = np.asarray(spot_key, dtype=np.float64)
    qq = np.asarray(image_query, dtype=np.float64)
    kk = kk / np.maximum(np.linalg.norm(kk, axis=1, keepdims=True), 1e-12)
    qq = qq / np.maximum(np.linalg.norm(qq, axis=1, keepdims=True), 1e-12)
    return qq @ kk.T

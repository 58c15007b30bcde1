#!/usr/bin/env python3
"""Generate ``retrieval_*.npz`` by running the REFERENCE's own retrieval code.

Runs only in the build container (needs /root/reference).  The reference's ``evel_*.py`` are scripts whose
module level reads datasets from absolute paths, so they cannot be imported.  This generator parses each
script with ``ast`` and executes, unmodified and in memory, exactly two pieces of it:

  * the ``find_matches`` function definition, and
  * the ``for i in range(indices.shape[0])`` weighting loop of the evaluation section,

on the procedural inputs of ``mclstexp_amd.synth.make_retrieval_case`` (regenerated from the formula by the
tests, never stored).  Only the reference's OUTPUTS are stored.

    python tests/golden/gen_retrieval_goldens.py        # writes tests/golden/retrieval_*.npz
"""
import ast
import contextlib
import io
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from mclstexp_amd import synth  # noqa: E402

REF = "/root/reference"

CASES = [
    # name, script, top_k, N, Q, G, seed, duplicates
    dict(name="her2st", script="evel_her2st.py", top_k=200, N=3000, Q=48, G=171, seed=1, duplicates=0),
    dict(name="cscc", script="evel_cscc.py", top_k=600, N=2500, Q=40, G=171, seed=2, duplicates=0),
    dict(name="visium", script="evel_visium.py", top_k=200, N=4100, Q=33, G=257, seed=3, duplicates=0),
    dict(name="small_k1", script="evel_visium.py", top_k=1, N=257, Q=5, G=19, seed=4, duplicates=0),
]


def lift(script: str):
    """(find_matches function object, compiled weighting loop) taken from the reference script."""
    src = open(os.path.join(REF, script), encoding="utf-8").read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "find_matches")
    loops = [n for n in ast.walk(tree) if isinstance(n, ast.For) and isinstance(n.iter, ast.Call)
             and "indices.shape[0]" in ast.unparse(n.iter) and "np.average" in ast.unparse(n)]
    assert len(loops) == 1, (script, len(loops))
    ns = {"torch": torch, "F": F, "np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), script, "exec"), ns)
    loop = compile(ast.Module(body=[loops[0]], type_ignores=[]), script, "exec")
    return ns["find_matches"], loop


def main():
    for c in CASES:
        find_matches, loop = lift(c["script"])
        d = synth.make_retrieval_case(c["N"], c["Q"], 256, c["G"], seed=c["seed"], duplicates=c["duplicates"])
        spot_key, image_query, expression_key = d["spot_key"], d["image_query"], d["expression_key"]
        with contextlib.redirect_stdout(io.StringIO()):
            res = find_matches(spot_key, image_query, top_k=c["top_k"])
        values, indices = res if isinstance(res, tuple) else (None, res)
        if indices.ndim == 1:   # top_k == 1 on a single query would squeeze; our cases keep Q > 1
            indices = indices[:, None]
        ns = {"np": np, "indices": indices, "spot_key": spot_key, "image_query": image_query,
              "expression_key": expression_key,
              "matched_spot_embeddings_pred": np.zeros((indices.shape[0], spot_key.shape[1])),
              "matched_spot_expression_pred": np.zeros((indices.shape[0], expression_key.shape[1]))}
        exec(loop, ns)
        out = dict(indices=indices.astype(np.int64),
                   emb_pred=ns["matched_spot_embeddings_pred"], expr_pred=ns["matched_spot_expression_pred"],
                   meta=np.array([c["N"], c["Q"], c["G"], c["top_k"], c["seed"], c["duplicates"]], dtype=np.int64))
        if values is not None:
            out["values"] = values
        np.savez_compressed(os.path.join(HERE, f"retrieval_{c['name']}.npz"), **out)
        print(c["name"], indices.shape, ns["matched_spot_expression_pred"].shape,
              "values" if values is not None else "")


if __name__ == "__main__":
    main()

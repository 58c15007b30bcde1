#!/usr/bin/env python3
"""Generate ``bleep_loss.npz`` by running the REFERENCE's own BLEEP soft-target CLIP loss.

Runs only in the build container (needs /root/reference).  ``baselines/Bleep/models.py`` cannot be imported (its
imports need timm and a config module with dataset paths); this generator parses it with ``ast`` and executes,
unmodified and in memory, the ``cross_entropy`` function and the loss section of ``CLIPModel.forward`` /
``CLIPModel_ViT.forward`` (the statements from ``logits = ...`` to the return) on procedural embeddings.  Only the
reference's OUTPUTS (loss and autograd gradients) are stored.

    python tests/golden/gen_bleep_goldens.py        # writes tests/golden/bleep_loss.npz
"""
import ast
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mclstexp_amd import synth  # noqa: E402

SRC = "/root/reference/baselines/Bleep/models.py"
CASES = [dict(name="clip_b8", cls="CLIPModel", B=8, T=1.0), dict(name="clip_b33_t07", cls="CLIPModel", B=33, T=0.7),
         dict(name="vit_b16_t05", cls="CLIPModel_ViT", B=16, T=0.5)]


def lift():
    tree = ast.parse(open(SRC, encoding="utf-8").read())
    ns = {"torch": torch, "nn": nn, "F": F}
    ce = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "cross_entropy")
    exec(compile(ast.Module(body=[ce], type_ignores=[]), SRC, "exec"), ns)
    fns = {}
    for cls in ("CLIPModel", "CLIPModel_ViT"):
        c = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
        fwd = next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == "forward")
        start = next(i for i, st in enumerate(fwd.body) if isinstance(st, ast.Assign)
                     and getattr(st.targets[0], "id", "") == "logits")
        f = ast.FunctionDef(name="loss_section", args=ast.arguments(
            posonlyargs=[], args=[ast.arg("self"), ast.arg("spot_embeddings"), ast.arg("image_embeddings")],
            kwonlyargs=[], kw_defaults=[], defaults=[]), body=fwd.body[start:], decorator_list=[])
        mod = ast.fix_missing_locations(ast.Module(body=[f], type_ignores=[]))
        exec(compile(mod, SRC, "exec"), ns)
        fns[cls] = ns["loss_section"]
    return fns


from helpers import bleep_embeddings as embeddings  # noqa: E402


def main():
    fns = lift()
    out = {}
    for i, c in enumerate(CASES):
        es, ei = embeddings(c["B"], i)
        es.requires_grad_(True); ei.requires_grad_(True)
        loss = fns[c["cls"]](types.SimpleNamespace(temperature=c["T"]), es, ei)
        loss.backward()
        out[c["name"] + ".loss"] = np.float64(loss.item())
        out[c["name"] + ".d_es"] = es.grad.numpy()
        out[c["name"] + ".d_ei"] = ei.grad.numpy()
        out[c["name"] + ".meta"] = np.array([c["B"], i, int(c["cls"] == "CLIPModel_ViT")], dtype=np.int64)
        out[c["name"] + ".T"] = np.float64(c["T"])
        print(c["name"], loss.item())
    np.savez_compressed(os.path.join(HERE, "bleep_loss.npz"), **out)


if __name__ == "__main__":
    main()

"""Builds libmclstexp_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m mclstexp_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libmclstexp_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "mclstexp_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(BUILD, os.path.basename(src)[:-4] + ".o")
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= _deps_mtime()):
        return obj
    cmd = ["hipcc", *FLAGS, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    if shutil.which("hipcc") is None:
        raise RuntimeError("hipcc not found on PATH; cannot build libmclstexp_hip.so")
    os.makedirs(BUILD, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = ["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"built {LIB} from {len(objs)} objects")
    elif verbose:
        print(f"{LIB} up to date")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

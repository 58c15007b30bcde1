"""Single-node multi-process launcher: one worker process per GPU (rank), torchrun-style environment.

``python bench.py --gpus N`` (and ``python -m mclstexp_amd.train --gpus N``) call ``spawn_workers`` from a
parent process that has made NO GPU call: the workers are fresh interpreters started with ``subprocess`` (never
``os.exec*`` from a process that has touched HIP), each with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT set, exactly what ``python -m torch.distributed.run --nproc-per-node N`` would export -- so the same
script works under either launcher.  Rank 0's stdout is relayed to the parent's stdout (the one JSON line of
bench.py); every rank's stderr goes to the parent's stderr.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time
from typing import List, Optional, Sequence


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def under_launcher() -> bool:
    """True inside a worker (torchrun or spawn_workers already set the rendezvous environment)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def spawn_workers(script: str, argv: Sequence[str], nproc: int, env_extra: Optional[dict] = None,
                  timeout: Optional[float] = None) -> int:
    """Runs ``python script *argv`` as ``nproc`` ranks on this node; returns the worst exit code."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    port = free_port()
    base = dict(os.environ)
    base.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "WORLD_SIZE": str(nproc),
                 "LOCAL_WORLD_SIZE": str(nproc), "HSA_ENABLE_IPC_MODE_LEGACY": base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    if env_extra:
        base.update({k: str(v) for k, v in env_extra.items()})
    procs: List[subprocess.Popen] = []
    relays: List[threading.Thread] = []

    def relay(stream, sink, prefix: str) -> None:
        for line in iter(stream.readline, ""):
            sink.write(prefix + line)
            sink.flush()
        stream.close()

    for r in range(nproc):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        p = subprocess.Popen([sys.executable, script, *argv], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True, bufsize=1)
        procs.append(p)
        # stdout: only rank 0 speaks (bench.py prints its JSON line there); other ranks' stdout is folded into stderr
        relays.append(threading.Thread(target=relay, args=(p.stdout, sys.stdout if r == 0 else sys.stderr,
                                                           "" if r == 0 else f"[rank {r}] "), daemon=True))
        relays.append(threading.Thread(target=relay, args=(p.stderr, sys.stderr, "" if r == 0 else f"[rank {r}] "),
                                       daemon=True))
    for t in relays:
        t.start()
    rc = 0
    t_end = None if timeout is None else time.monotonic() + timeout
    try:
        while any(p.poll() is None for p in procs):
            bad = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
            if bad or (t_end is not None and time.monotonic() > t_end):
                rc = bad[0] if bad else 124
                for q in procs:                      # one rank failed: the others would hang in a collective
                    if q.poll() is None:
                        q.terminate()
                break
            time.sleep(0.05)
        for p in procs:
            try:
                code = p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                code = 124
            rc = rc or code
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    for t in relays:
        t.join(timeout=5)
    return rc

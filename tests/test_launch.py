"""bench.py --gpus N must launch itself: the parent makes no GPU call, spawns N fresh
worker processes with the torchrun environment, relays rank 0's single JSON line and returns the workers' worst
exit code.  Exercised here on CPU (gloo) through bench.py's --launch_check mode."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_bench_self_launches_two_ranks():
    r = _run(["--gpus", "2", "--launch_check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]      # (gloo itself chats on stdout; RCCL does not)
    assert len(lines) == 1, r.stdout                        # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "world": 2, "sum": 3.0}


def test_bench_world_mismatch_is_an_error():
    # started by a launcher with 2 ranks but told --gpus 4: refuse instead of silently benchmarking the wrong job
    r = _run(["--gpus", "4", "--launch_check"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2",
                                                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29655"}, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_spawn_workers_propagates_failure(tmp_path):
    from mclstexp_amd import launch
    script = tmp_path / "w.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "if r == 1: sys.exit(7)\n"
                      "time.sleep(30)\n")                      # the other ranks 'hang in a collective'
    import time
    t0 = time.time()
    rc = launch.spawn_workers(str(script), [], 3)
    assert rc == 7 and time.time() - t0 < 20                  # the failed rank takes the job down promptly

#!/usr/bin/env python3
"""Per-launch durations of selected kernels inside ONE steady training step of a rocprofv3 kernel trace (rocpd SQLite):
the launches are listed in issue order, so the 58 dense layers of DenseNet-121 can be read off block by block.

    python tools/layer_times.py DB substring [substring ...]
"""
import sqlite3
import sys

db, subs = sys.argv[1], sys.argv[2:]
c = sqlite3.connect(db)
rows = c.execute('select name, start, "end" from kernels order by start').fetchall()
marks = [i for i, r in enumerate(rows) if "adam_consts_kernel" in r[0]]
lo, hi = marks[-2], marks[-1]            # one complete step (one adam_consts_kernel launch per step)
step = rows[lo + 1: hi + 1]
print(f"step wall {(step[-1][2] - step[0][1]) / 1e3:.1f} us, {len(step)} launches")
for s in subs:
    d = [round((e - b) / 1e3, 1) for n, b, e in step if s in n]
    print(s, len(d), "launches, total", round(sum(d), 1), "us")
    print("  ", d)

#!/usr/bin/env python3
"""Timeline of the LAST training step from a rocprofv3 rocpd kernel trace: start offset, duration, queue (stream) and
kernel name per launch, plus the time each queue is busy and the union:
    python tools/trace_timeline.py DB OUT.txt   (step delimited by adam_table_kernel launches, 2 per step)"""
import re
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = f'select name, start, "end", {qcol} from kernels order by start' if qcol else \
    'select name, start, "end", 0 from kernels order by start'
rows = c.execute(sel).fetchall()
marks = [i for i, r in enumerate(rows) if "adam_table_kernel" in r[0]]
i0, i1 = marks[-3] + 1, marks[-1] + 1
step = rows[i0:i1]
t0 = step[0][1]
busy = {}
with open(out, "w") as f:
    for n, s, e, q in step:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void\s+", "", n)
        busy[q] = busy.get(q, 0) + (e - s)
        f.write(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} us  q{q}  {n[:100]}\n")
    f.write("\n")
    for q, b in sorted(busy.items(), key=lambda kv: -kv[1]):
        f.write(f"queue {q}: busy {b / 1e3:.1f} us\n")
    f.write(f"step span {(step[-1][2] - t0) / 1e3:.1f} us, {len(step)} launches\n")
print(len(step), "launches; columns:", cols)

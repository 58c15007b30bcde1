#!/usr/bin/env python3
"""GPU idle time inside the steady-state training steps of a rocprofv3 rocpd kernel trace: union of kernel intervals
vs wall clock, and which kernels the idle gaps follow.   python tools/trace_gaps.py DB [--steady K]"""
import collections
import re
import sqlite3
import sys

db = sys.argv[1]
K = int(sys.argv[sys.argv.index("--steady") + 1]) if "--steady" in sys.argv else 6
c = sqlite3.connect(db)
rows = c.execute('select name, start, "end" from kernels order by start').fetchall()
marks = [i for i, r in enumerate(rows) if "adam_consts_kernel" in r[0]]
i0, i1 = marks[-K - 1] + 1, marks[-1] + 1
rows = rows[i0:i1]
t0, t1 = rows[0][1], max(r[2] for r in rows)
busy_end = rows[0][1]
idle = 0
gaps = collections.defaultdict(lambda: [0, 0.0])
last = rows[0][0]
hist = collections.Counter()
pairs = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    if s > busy_end:
        g = (s - busy_end) / 1e3
        idle += s - busy_end
        key = re.sub(r"\(anonymous namespace\)::|^void\s+", "", last).split("(")[0][:50]
        gaps[key][0] += 1
        gaps[key][1] += g
        nxt = re.sub(r"\(anonymous namespace\)::|^void\s+", "", n).split("(")[0][:40]
        pairs[(key[:40], nxt)][0] += 1
        pairs[(key[:40], nxt)][1] += g
        hist[min(int(g), 20)] += 1
    if e > busy_end:
        busy_end, last = e, n
print(f"{K} steps: wall {(t1 - t0) / 1e3 / K:.1f} us/step, kernel-sum {sum(e - s for _, s, e in rows) / 1e3 / K:.1f}, "
      f"idle (no kernel resident) {idle / 1e3 / K:.1f} us/step in {sum(v[0] for v in gaps.values()) / K:.0f} gaps/step")
print("gap length histogram (us: count/step):", {k: round(v / K, 1) for k, v in sorted(hist.items())})
for k, (n, t) in sorted(gaps.items(), key=lambda x: -x[1][1])[:14]:
    print(f"  after {k:52s} {n / K:6.1f} gaps/step {t / K:8.1f} us/step  avg {t / n:5.2f} us")
print("gaps by (kernel that ended last, kernel that started next):")
for (a, b), (n, t) in sorted(pairs.items(), key=lambda x: -x[1][1])[:12]:
    print(f"  {a:42s} -> {b:42s} {n / K:6.1f} gaps/step {t / K:8.1f} us/step  avg {t / n:5.2f} us")

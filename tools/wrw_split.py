#!/usr/bin/env python3
"""Per-shape medians of wrw_partial_kernel / wrw_merge_kernel from a rocprofv3 kernel trace of
`tools/bench_dense_layer.py --only bn1_wrw` (8 shapes, equal launch counts)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute('select name, start, "end" from kernels order by start').fetchall()
seq = [(n, (e - s) / 1e3) for n, s, e in rows if "wrw_partial" in n or "wrw_merge" in n]
per = len(seq) // 8
shapes = ["401k,64", "401k,224", "100k,128", "100k,480", "25k,256", "25k,992", "6k,512", "6k,992"]
for i in range(8):
    blk = seq[i * per:(i + 1) * per]
    pa = sorted(d for n, d in blk if "partial" in n)
    me = sorted(d for n, d in blk if "merge" in n)
    print(f"{shapes[i]:10s} partial {pa[len(pa) // 2]:6.1f} us   merge {me[len(me) // 2]:6.1f} us")

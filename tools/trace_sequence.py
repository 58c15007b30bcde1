#!/usr/bin/env python3
"""Dump the kernel launch sequence of the LAST training step from a rocprofv3 rocpd database:
    python tools/trace_sequence.py DB OUT.txt   (step delimited by adam_table_kernel launches)"""
import re
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute('select name, start, "end" from kernels order by start').fetchall()
marks = [i for i, r in enumerate(rows) if "adam_table_kernel" in r[0]]
i0, i1 = marks[-3] + 1, marks[-1] + 1
with open(out, "w") as f:
    for n, s, e in rows[i0:i1]:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void\s+", "", n)
        f.write(f"{(e - s) / 1e3:9.1f} us  {n[:110]}\n")
print(i1 - i0, "launches in the last step")

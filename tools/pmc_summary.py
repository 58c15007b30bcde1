#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter CSVs:  python tools/pmc_summary.py DIR [substring ...]"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
subs = sys.argv[2:] or ["strip_kernel"]
res = collections.defaultdict(dict)
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in subs):
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", k)
        name = re.sub(r"^void\s+", "", name).split("(")[0][-60:]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in agg.items():
        for c, v in cs.items():
            res[name][c] = (sum(v) / len(v), len(v))
for name, cs in res.items():
    print(name)
    for c, (v, n) in sorted(cs.items()):
        print(f"   {c:28s} {v:16.0f}   (n={n})")

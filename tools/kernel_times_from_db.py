#!/usr/bin/env python3
"""Per-kernel launch durations (median / min, by kernel and grid) from a rocprofv3 rocpd database (--kernel-trace)."""
import collections
import glob
import sqlite3
import sys


def main():
    path = sys.argv[1]
    if not path.endswith(".db"):
        path = sorted(glob.glob(path + "/**/*.db", recursive=True))[0]
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    agg = collections.defaultdict(list)
    for n, s, e, gx, gy in c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y from {kd} d join {ks} s "
                                     "on d.kernel_id = s.id"):
        agg[(n.split("(")[0][-64:], gx, gy)].append(e - s)
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
        v = sorted(v)
        print(f"{k[0]:64s} grid {k[1]:6d} x {k[2]:3d}  launches {len(v):6d}  median {v[len(v) // 2] / 1e3:8.2f} us  min {v[0] / 1e3:8.2f} us")


if __name__ == "__main__":
    main()

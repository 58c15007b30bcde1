#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace as CSV: per-kernel calls / total /
average -- the content of `rocprofv3 --kernel-trace --stats`.

    python tools/rocpd_stats.py DB OUT.csv [--steps N]            # whole process, per-step columns = /N
    python tools/rocpd_stats.py DB OUT.csv --steady K             # only the LAST K training steps, delimited by
                                                                  # adam_table_kernel launches (2 per step), which
                                                                  # drops MIOpen's first-step solver search
"""
import argparse
import csv
import re
import sqlite3


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void\s+", "", name)
    return name if len(name) <= 150 else name[:147] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("out")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--steady", type=int, default=None)
    ap.add_argument("--marker", default="adam_consts_kernel")
    ap.add_argument("--per_step", type=int, default=1, help="marker launches per step")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute('select name, start, "end" from kernels order by start').fetchall()
    steps = a.steps
    if a.steady:
        marks = [r for r in rows if a.marker in r[0]]
        need = a.steady * a.per_step
        if len(marks) <= need:
            raise SystemExit(f"only {len(marks)} {a.marker} launches; cannot isolate the last {a.steady} steps")
        t0 = marks[-need - 1][2]          # end of the last marker launch before the window
        t1 = marks[-1][2]
        rows = [r for r in rows if r[1] >= t0 and r[2] <= t1]
        steps = a.steady
        wall = (t1 - t0) / 1e3
    agg = {}
    for n, s, e in rows:
        k = agg.setdefault(n, [0, 0.0])
        k[0] += 1
        k[1] += (e - s) / 1e3
    tot = sum(v[1] for v in agg.values())
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(a.out, "w", newline="") as f:
        w = csv.writer(f)
        hdr = ["kernel", "calls", "total_us", "avg_us", "percent"] + (["calls_per_step", "us_per_step"] if steps else [])
        w.writerow(hdr)
        for n, (calls, total) in items:
            r = [short(n), calls, round(total, 1), round(total / calls, 3), round(100 * total / tot, 3)]
            if steps:
                r += [round(calls / steps, 2), round(total / steps, 1)]
            w.writerow(r)
        ncalls = sum(v[0] for v in agg.values())
        w.writerow(["TOTAL", ncalls, round(tot, 1), "", 100.0] +
                   ([round(ncalls / steps, 1), round(tot / steps, 1)] if steps else []))
        if a.steady:
            w.writerow(["WALL_us_per_step (window)", "", "", "", "", "", round(wall / steps, 1)])
    msg = f"{len(items)} kernels, {tot/1e3:.1f} ms kernel time"
    if steps:
        msg += f", {tot/1e3/steps:.2f} ms/step over {steps} steps"
    if a.steady:
        msg += f", wall {wall/1e3/steps:.2f} ms/step"
    print(msg)


if __name__ == "__main__":
    main()

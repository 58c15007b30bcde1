#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace: per-kernel calls / total / average,
the same content as `rocprofv3 --kernel-trace --stats` prints, as CSV.

    python tools/rocpd_stats.py gpurun_out/prof/r1_results.db profiles/r1_bench_kernel_stats.csv [steps]
"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void\s+", "", name)
    if len(name) > 150:
        name = name[:147] + "..."
    return name


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        hdr = ["kernel", "calls", "total_us", "avg_us", "percent"]
        if steps:
            hdr += ["calls_per_step", "us_per_step"]
        w.writerow(hdr)
        for n, calls, total, avg, pct in rows:
            r = [short(n), calls, round(total, 1), round(avg, 3), round(pct, 3)]
            if steps:
                r += [round(calls / steps, 2), round(total / steps, 1)]
            w.writerow(r)
        w.writerow(["TOTAL", sum(r[1] for r in rows), round(tot, 1), "", 100.0] +
                   ([round(sum(r[1] for r in rows) / steps, 1), round(tot / steps, 1)] if steps else []))
    print(f"{len(rows)} kernels, {tot/1e3:.1f} ms total" + (f", {tot/1e3/steps:.2f} ms/step" if steps else ""))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.x rocpd SQLite) kernel trace: per-kernel calls / total / average
duration, like `--stats` CSV.  Usage: rocpd_summary.py results.db [> profiles/xxx.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("imx::(anonymous namespace)::", "")
    name = re.sub(r"\((imx::)?\w+Args.*$", "", name)       # drop the argument list
    name = re.sub(r"\(float const\*.*$", "", name)
    return name[:70]


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, (end - start) from kernels").fetchall()
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += dur
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{short(name):70s} {n:7d} {ns / 1e6:10.3f} {ns / n / 1e3:10.2f} {100.0 * ns / tot:6.2f}")
    print(f"{'TOTAL':70s} {len(rows):7d} {tot / 1e6:10.3f}")


if __name__ == "__main__":
    main(sys.argv[1])

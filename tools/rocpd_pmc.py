#!/usr/bin/env python
"""Per-kernel PMC counter averages from a rocprofv3 rocpd database (counters_collection view)."""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name).replace("imx::(anonymous namespace)::", "")
    name = re.sub(r"\((imx::)?\w+Args.*$", "", name)
    return re.sub(r"\(float const\*.*$", "", name)[:44]


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select kernel_name, dispatch_id, counter_name, value, (end-start), grid_size, workgroup_size "
                     "from counters_collection").fetchall()
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for k, d, cn, v, dt, g, wg in rows:
        per[short(k)][cn].append(v)
        dur[short(k)][d] = dt
    names = sorted({cn for k in per for cn in per[k]})
    print(f"{'kernel':44s} {'n':>4s} {'avg_us':>9s} " + " ".join(f"{n[-18:]:>18s}" for n in names))
    for k in sorted(per, key=lambda k: -sum(dur[k].values())):
        n = len(dur[k])
        print(f"{k:44s} {n:4d} {sum(dur[k].values()) / n / 1e3:9.1f} " +
              " ".join(f"{sum(per[k][cn]) / max(1, len(per[k][cn])):18.4g}" for cn in names))


if __name__ == "__main__":
    main(sys.argv[1])

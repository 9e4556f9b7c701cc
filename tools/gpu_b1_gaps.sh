#!/bin/bash
# single-pair step: sum of kernel durations (rocprofv3 kernel trace) vs wall time per iteration -> how much of the latency is gaps
R=$(pwd); export TMPDIR=/tmp
out=$R/gpurun_out; rm -rf $out/prof_tmp; mkdir -p $out/prof_tmp
(cd /tmp && rocprofv3 --kernel-trace -d $out/prof_tmp -- python $R/tools/run_pairs.py --pairs 1 --iters 200 > $out/prof_tmp/run.log 2>&1)
db=$(find $out/prof_tmp -name "*.db" | head -1)
python - "$db" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
rows = c.execute(f"select start, end from {view[0]} order by start").fetchall()
n = len(rows)
# steady state: last 100 iterations' worth of launches
per = n // 200
tail = rows[-100 * per:]
busy = sum(e - s for s, e in tail)
wall = tail[-1][1] - tail[0][0]
gaps = sorted((tail[i + 1][0] - tail[i][1]) for i in range(len(tail) - 1))
print(f"launches per pair {per}; per pair: kernel time {busy / 100 / 1e3:.1f} us, wall {wall / 100 / 1e3:.1f} us, gaps {(wall - busy) / 100 / 1e3:.1f} us; "
      f"median gap {gaps[len(gaps) // 2] / 1e3:.2f} us, p90 {gaps[int(len(gaps) * .9)] / 1e3:.2f} us")
PY
rm -rf $out/prof_tmp

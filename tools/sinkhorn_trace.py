#!/usr/bin/env python
"""Workgroup lives of the Sinkhorn slab kernel: needs a library whose sg_misc.hip was compiled with -DSK_TRACE
(put `#define SK_TRACE 1` at its top and `make`; the product build carries no stamps).  Runs one fused call of --pairs pairs with the
debug taps on, reads the `sk_trace` tap (last iteration: s_memrealtime at entry / after each slab / at exit, HW_ID, XCC_ID per
workgroup) and prints when workgroups start, how long they live, how long each slab takes and how many are alive over the launch.
    python tools/sinkhorn_trace.py --workload c5 --pairs 8"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench  # noqa: E402
from image_matching_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=8)
ap.add_argument("--workload", default="c5")
a = ap.parse_args()
wl = bench.WORKLOADS[a.workload]
m, *_ = bench.build_matching(wl, torch.device("cuda", 0))
eng = m._shared.get_engine([0, 1])
ims = [synth.synth_pair(i, wl["H"], wl["W"]) for i in range(a.pairs)]
i0 = torch.from_numpy(np.stack([p[0] for p in ims]))[:, None].cuda()
i1 = torch.from_numpy(np.stack([p[1] for p in ims]))[:, None].cuda()
m.match_batch(i0, i1)
torch.cuda.synchronize()
eng.set_debug(True)
m.match_batch(i0, i1)
torch.cuda.synchronize()
raw = eng.fetch("sk_trace")
eng.set_debug(False)
t = np.ascontiguousarray(raw).view(np.uint64).reshape(-1, 8)
t = t[t[:, 0] > 0]
if not len(t):
    sys.exit("no stamps: the library was not built with -DSK_TRACE")
tick = 0.01                                     # s_memrealtime: 100 MHz -> microseconds
start = (t[:, 0] - t[:, 0].min()).astype(np.float64) * tick
end = (t[:, 5] - t[:, 0].min()).astype(np.float64) * tick
life = end - start
hw, xcc = (t[:, 6] & 0xffffffff).astype(np.int64), (t[:, 6] >> 32).astype(np.int64) & 0xf
cu, se = (hw >> 8) & 0xf, (hw >> 13) & 0x7
q = lambda x: " / ".join(f"{v:.1f}" for v in np.percentile(x, [0, 10, 50, 90, 100]))
print(f"{a.workload} x {a.pairs} pairs: {len(t)} workgroups stamped; launch span {end.max():.1f} us (first entry to last exit)")
print(f"  entry after the first one, us (min / p10 / median / p90 / max): {q(start)}")
print(f"  life (entry to exit), us:                                      {q(life)}")
prev = t[:, 0]
for g in range(4):
    have = t[:, 1 + g] > 0
    if have.any():
        d = (t[have, 1 + g] - prev[have]).astype(np.float64) * tick
        print(f"  slab {g} ({have.sum()} workgroups), us:                                   {q(d)}")
        prev = np.where(have, t[:, 1 + g], prev)
late = start > 2.0
print(f"  workgroups entering more than 2 us after the first: {late.sum()} (their lives: {q(life[late]) if late.any() else '-'})")
edges = np.arange(0.0, end.max() + 1.0, max(1.0, end.max() / 24))
alive = [(int(((start <= e) & (end > e)).sum())) for e in edges]
print("  alive at t (us): " + "  ".join(f"{e:.0f}:{n}" for e, n in zip(edges, alive)))
for x in sorted(set(xcc.tolist())):
    sel = xcc == x
    print(f"  XCC {x}: {sel.sum()} workgroups, entry median {np.median(start[sel]):.1f} us, life median {np.median(life[sel]):.1f}, last exit {end[sel].max():.1f}; "
          f"distinct (SE, CU) pairs {len(set(zip(se[sel].tolist(), cu[sel].tolist())))}")

#!/usr/bin/env python
"""Time the Sinkhorn launches of the fused pair path for a few settings of the handle option "sinkhorn_group" (slabs per workgroup), same process,
same box: tools/sinkhorn_time.py --workload c3 --pairs 64 --groups 1 2 4 8.  Per setting: HIP events around every launch
(imx_set_timing), ms per step summed over the 'sinkhorn' rows, median of --reps steps."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench  # noqa: E402
from image_matching_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=64)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--workload", default="c3")
ap.add_argument("--groups", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--prefetch", type=int, nargs="+", default=[None], help="\"sinkhorn_prefetch\" settings to cross with --groups (0 / 1)")
a = ap.parse_args()
wl = bench.WORKLOADS[a.workload]
m, *_ = bench.build_matching(wl, torch.device("cuda", 0))
eng = m._shared.get_engine([0, 1])
ims = [synth.synth_pair(i, wl["H"], wl["W"]) for i in range(a.pairs)]
i0 = torch.from_numpy(np.stack([p[0] for p in ims]))[:, None].cuda()
i1 = torch.from_numpy(np.stack([p[1] for p in ims]))[:, None].cuda()
m.match_batch(i0, i1)
torch.cuda.synchronize()
for rnd in range(2):
    for G, PF in [(g, p) for g in a.groups for p in a.prefetch]:
        eng.set_option("sinkhorn_group", G)
        if PF is not None:
            eng.set_option("sinkhorn_prefetch", PF)
        m.match_batch(i0, i1)
        vals, tot = [], []
        for _ in range(a.reps):
            eng.timing_reset()
            eng.set_timing(True)
            m.match_batch(i0, i1)
            torch.cuda.synchronize()
            rows = {r[0]: r for r in eng.timing_report()}
            eng.set_timing(False)
            vals.append(rows["sinkhorn"][2])
            tot.append(sum(r[2] for r in rows.values()))
        print(f"{a.workload} pairs {a.pairs} G={G} prefetch={PF}: sinkhorn {np.median(vals):.3f} ms per step (sum of kernels {np.median(tot):.2f} ms)", flush=True)

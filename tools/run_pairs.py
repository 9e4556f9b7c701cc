#!/usr/bin/env python
"""Run the fused pair path a few times (profiling target for rocprofv3)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench  # noqa: E402
from image_matching_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=16)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--workload", default="c3")
a = ap.parse_args()
wl = bench.WORKLOADS[a.workload]
m, *_ = bench.build_matching(wl, torch.device("cuda", 0))
ims = [synth.synth_pair(i, wl["H"], wl["W"]) for i in range(a.pairs)]
i0 = torch.from_numpy(np.stack([p[0] for p in ims]))[:, None].cuda()
i1 = torch.from_numpy(np.stack([p[1] for p in ims]))[:, None].cuda()
for _ in range(a.iters):
    out = m.match_batch(i0, i1)
torch.cuda.synchronize()
print("counts ok:", bool((out["counts0"] == wl["K"]).all()), "matches/pair:", float((out["matches0"] > -1).sum()) / a.pairs)

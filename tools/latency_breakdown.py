#!/usr/bin/env python
"""Per-kernel time of ONE pair through the fused call (BASELINE configs[2]): HIP events per launch group (imx_set_timing)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench
from image_matching_amd import synth
wl = bench.WORKLOADS["c3"]
m, *_ = bench.build_matching(wl, torch.device("cuda", 0))
im0, im1 = synth.synth_pair(7, wl["H"], wl["W"])
i0, i1 = torch.from_numpy(im0)[None, None].cuda(), torch.from_numpy(im1)[None, None].cuda()
for _ in range(5):
    m.match_batch(i0, i1)
eng = m._shared.engine
eng.timing_reset(); eng.set_timing(True)
N = 20
for _ in range(N):
    m.match_batch(i0, i1)
rows = eng.timing_report(forms=True)
eng.set_timing(False)
tot = sum(r[2] for r in rows) / N
print(f"sum of kernel groups: {tot * 1e3:.0f} us per pair")
for name, launches, ms, form in sorted(rows, key=lambda r: -r[2]):
    print(f"  {name:16s} {launches // N:4d} launches  {ms / N * 1e3:7.1f} us  ({ms / launches * 1e3:6.1f} us each)  {form}")

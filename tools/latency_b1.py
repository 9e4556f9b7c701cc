#!/usr/bin/env python
"""Single-pair latency of the drop-in Matching.forward (the reference CLI's batch-1 loop) and of the fused batch path at B = 1."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from image_matching_amd import synth
wl = bench.WORKLOADS["c3"]
m, *_ = bench.build_matching(wl, torch.device("cuda", 0))
im0, im1 = synth.synth_pair(0, wl["H"], wl["W"])
x0 = torch.from_numpy(im0)[None, None].cuda(); x1 = torch.from_numpy(im1)[None, None].cuda()
for name, fn in (("Matching.forward (dict API)", lambda: m({"image0": x0, "image1": x1})), ("match_batch, B=1", lambda: m.match_batch(x0, x1))):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
    print(f"{name}: {dt*1e3:.3f} ms per pair")
eng = m._shared.engine
eng.timing_reset(); eng.set_timing(True)
for _ in range(10): m.match_batch(x0, x1)
torch.cuda.synchronize()
rows = eng.timing_report(); eng.set_timing(False)
print("sum of kernel times per pair: %.3f ms over %d launches" % (sum(r[2] for r in rows) / 10, sum(r[1] for r in rows) / 10))
for r in sorted(rows, key=lambda r: -r[2]):
    print(f"  {r[0]:16s} launches/pair {r[1] / 10:6.1f}  ms/pair {r[2] / 10:7.4f}  us/launch {1e3 * r[2] / r[1]:7.2f}")

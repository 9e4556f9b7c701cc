#!/usr/bin/env python
"""Chunk stamps of gemm_h2 (a -DGH2_TRACE build of gemm_h2.hip): per chunk of the first workgroups, s_memtime at entry (after the
previous chunk's barrier) | fragments arrived | MFMAs and split issued | after the barrier.  tools/gemm_h2_trace.py --workload c5 --pairs 8"""
import argparse, os, sys
import numpy as np, torch
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
eng.set_debug(True)
m.match_batch(i0, i1)
torch.cuda.synchronize()
for name in ("qkv_proj", "gnn_mlp1", "gnn_mlp2"):
    try:
        raw = eng.fetch("gh2_trace_" + name).view(np.uint64).reshape(64, 128).astype(np.int64)
        t = raw[:, :96].reshape(64, 24, 4)
        ep = raw[:, 96:].reshape(64, 4, 8)
    except Exception as e:
        print(name, "no tap", e); continue
    ok = t[:, :, 0] > 0
    if not ok.any():
        print(name, "no stamps (build without -DGH2_TRACE?)"); continue
    n = int(ok[0].sum())
    lvl2 = bool((t[:, :n, 1] > 0).any())
    if not lvl2:
        t[:, :n, 1] = t[:, :n, 0]
    d01 = (t[:, :n, 1] - t[:, :n, 0]); d12 = (t[:, :n, 2] - t[:, :n, 1]); d23 = (t[:, :n, 3] - t[:, :n, 2])
    per = np.diff(t[:, :n, 0], axis=1)
    print(f"{name}: {n} chunks stamped per workgroup (64 workgroups); s_memtime ticks (100 MHz constant clock x ? -- raw units)")
    print("  chunk period          median", np.median(per), " p10", np.percentile(per, 10), " p90", np.percentile(per, 90))
    print("  entry -> frags here   median", np.median(d01), " p90", np.percentile(d01, 90))
    print("  frags -> issued       median", np.median(d12), " p90", np.percentile(d12, 90))
    print("  issued -> barrier out median", np.median(d23), " p90", np.percentile(d23, 90))
    for k in range(4):
        if (ep[:, k, 0] > 0).any():
            e = ep[:, k]
            print(f"  epilogue {k}: entry -> stores issued median {np.median(e[:, 1] - e[:, 0])}  -> maxima done {np.median(e[:, 2] - e[:, 1])}  -> vmcnt(0) {np.median(e[:, 3] - e[:, 2])}   (p90 of the drain {np.percentile(e[:, 3] - e[:, 2], 90)})")
    print("  workgroup 0, chunk by chunk (entry delta | e->f | f->i | i->b):")
    for q in range(min(n - 1, 20)):
        print(f"    q={q:2d}  {per[0, q]:7d} | {d01[0, q]:6d} {d12[0, q]:6d} {d23[0, q]:6d}")

#!/usr/bin/env python
"""PCIe-inclusive C3 rate: decoded uint8 frames (960x1280, resize_scale 0.5 -> 480x640) start in pinned host memory and
go through IngestPipeline (async H2D + resize_u8_unit on a side stream) into the fused matcher.  Reported next to
bench.py's HBM-resident figure in DESIGN.md; it is never bench.py's `value`."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_matching_amd  # noqa: F401,E402
from image_matching_amd import hostops, synth  # noqa: E402
from image_matching_amd.ingest import IngestPipeline  # noqa: E402
from image_matching_amd.superglue.models.matching_test import Matching  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=0.5)
    ap.add_argument("--staged", action="store_true", help="frames already sit in the pinned staging buffers")
    a = ap.parse_args()
    tt = lambda sd: {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    kenc, iters, thr = synth.SG_CONFIGS[128]
    m = Matching({"superpoint": {"weights": None, "descriptor_dim": 128, "nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": 1024},
                  "superglue": {"weights": None, "descriptor_dim": 128, "keypoint_encoder": kenc, "sinkhorn_iterations": iters,
                                "match_threshold": thr}}).eval().to("cuda")
    m.superpoint.load_state_dict(tt(synth.make_superpoint_state_dict(128)))
    m.superglue.load_state_dict(tt(synth.make_superglue_state_dict(128)))
    Hs, Ws = int(round(480 / a.scale)), int(round(640 / a.scale))
    f0, f1 = [], []
    for i in range(a.pairs):
        im0, im1 = synth.synth_pair(1000 + i, 480, 640)
        f0.append(hostops.resize_linear_u8((im0 * 255).astype(np.uint8), (Ws, Hs)))
        f1.append(hostops.resize_linear_u8((im1 * 255).astype(np.uint8), (Ws, Hs)))
    eng = m._shared.get_engine([0, 1])
    p0, p1 = IngestPipeline(eng, a.pairs, (Hs, Ws), (480, 640)), IngestPipeline(eng, a.pairs, (Hs, Ws), (480, 640))

    def ship():
        if a.staged:                   # the decoder wrote into page-locked memory already: no extra host copy
            p0.staging(), p1.staging()
            return p0.submit_staged(a.pairs), p1.submit_staged(a.pairs)
        return p0.submit(f0), p1.submit(f1)

    def run(n):
        t = ship()
        for k in range(n):             # match(k) is enqueued before batch k+1 is staged: the copy overlaps the match
            m.match_batch(p0.take(t[0]), p1.take(t[1]))
            p0.release(t[0]), p1.release(t[1])
            t = ship() if k + 1 < n else None
        torch.cuda.synchronize()
    if a.staged:
        for p, f in ((p0, f0), (p1, f1)):
            for s in p.slots:
                s["pinned"].copy_(torch.from_numpy(np.stack(f)))
    run(a.warmup)
    t0 = time.perf_counter()
    run(a.steps)
    dt = time.perf_counter() - t0
    h2d = 2 * a.pairs * Hs * Ws * a.steps / dt / 1e9
    print(json.dumps({"metric": "image-pairs/sec, uint8 frames in pinned host memory (PCIe-inclusive)", "value": a.pairs * a.steps / dt,
                      "pairs_per_step": a.pairs, "steps": a.steps, "frame": [Hs, Ws], "resize_scale": a.scale, "staged": a.staged, "h2d_GBps": h2d,
                      "ms_per_step": dt / a.steps * 1e3}))


if __name__ == "__main__":
    main()

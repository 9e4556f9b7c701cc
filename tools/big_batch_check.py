#!/usr/bin/env python
"""Developer check (GPU box): very large batches -- do all copies of the same pair give the same bytes?  Catches 32-bit offset
overflows (S of 1100 C3 pairs is 4.6 GB; the first layer's output of 600 images 11.8 GB).
usage: python tools/big_batch_check.py [n_superglue_pairs] [n_matching_pairs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util  # noqa: E402
from image_matching_amd.superglue.models.matching_test import Matching  # noqa: E402

nsg = int(sys.argv[1]) if len(sys.argv) > 1 else 1100
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
d, K, H, W = 128, 1024, 480, 640
m = Matching({"superpoint": util.sp_config(d, K), "superglue": util.sg_config(d)}).eval().to("cuda")
m.superpoint.load_state_dict(util.sp_sd(d))
m.superglue.load_state_dict(util.sg_sd(d))
eng = m._shared.get_engine([0, 1])
eng.set_option("latency_forms", "off")
pairs = [util.pair(s, H, W) for s in (1000, 1001, 1002)]
i0 = torch.cat([p[0] for p in pairs]).cuda()
i1 = torch.cat([p[1] for p in pairs]).cuda()
base = m.match_batch(i0, i1, want_desc=True)
torch.cuda.synchronize()
# ---- Matching: nm pairs, copies of the three
idx = torch.arange(nm) % 3
out = m.match_batch(i0[idx], i1[idx], want_desc=True)
torch.cuda.synchronize()
bad = 0
for k in ("keypoints0", "keypoints1", "scores0", "descriptors1", "matches0", "matches1", "matching_scores0"):
    same = (out[k] == base[k][idx.to(out[k].device)]).flatten(1).all(1)
    bad += int((~same).sum())
    print(f"matching B={nm}: {k}: {int((~same).sum())} of {nm} pairs differ from the 3-pair call")
# ---- SuperGlue alone: nsg copies of pair 0's features
rep = lambda t: t[:1].expand(nsg, *t.shape[1:]).contiguous()
one = eng.superglue(base["keypoints0"][:1], base["scores0"][:1], base["descriptors0"][:1].transpose(1, 2), (1, 1, H, W),
                    base["keypoints1"][:1], base["scores1"][:1], base["descriptors1"][:1].transpose(1, 2), (1, 1, H, W))
big = eng.superglue(rep(base["keypoints0"]), rep(base["scores0"]), rep(base["descriptors0"]).transpose(1, 2), (1, 1, H, W),
                    rep(base["keypoints1"]), rep(base["scores1"]), rep(base["descriptors1"]).transpose(1, 2), (1, 1, H, W))
torch.cuda.synchronize()
for name, a, b in zip(("matches0", "matches1", "mscores0", "mscores1"), one, big):
    same = (b == a[:1]).flatten(1).all(1)
    bad += int((~same).sum())
    print(f"superglue B={nsg}: {name}: {int((~same).sum())} of {nsg} copies differ" + (f" (first at {int((~same).nonzero()[0])})" if not same.all() else ""))
print("RESULT", "ok" if bad == 0 else f"{bad} differing copies")

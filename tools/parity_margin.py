#!/usr/bin/env python
"""How much of the 1e-4 + 1e-4*|ref| tolerance do the shipped kernels use?  Dense SuperPoint stages against the reference
goldens (tests/golden/sp_small.npz, sp_ragged.npz), for the default kernels (Winograd on the bf16 pipe), the same arithmetic on the fp32 MFMA, and the direct form."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from tests import util
from image_matching_amd import _lib as L
from image_matching_amd.engine import Engine
for name in ("sp_small.npz", "sp_ragged.npz"):
    g = util.golden(name)
    H, W, seed, K = int(g["H"]), int(g["W"]), int(g["seed"]), int(g["max_keypoints"])
    eng = Engine(util.sp_config(128, K), util.sg_config(128), "cuda", 0, None)
    eng.load_state_dict(L.NET_SUPERPOINT, util.sp_sd(128))
    eng.set_debug(True)
    eng.superpoint(torch.cat(util.pair(seed, H, W)).cuda())
    nchw = lambda a: np.transpose(a, (0, 3, 1, 2))
    raw = nchw(eng.fetch("desc_raw"))
    got = {"x4": nchw(eng.fetch("x4")), "semi": nchw(eng.fetch("semi")), "desc": raw / np.linalg.norm(raw, axis=1, keepdims=True),
           "score_map": eng.fetch("score_map")}
    out = []
    for k, a in got.items():
        b = g[k].astype(np.float64)
        atol = 1e-5 if k == "score_map" else 1e-4
        r = (np.abs(a.astype(np.float64) - b) / (atol + 1e-4 * np.abs(b))).max()
        out.append("%%s %%.3f" %% (k, r))
    print(name, " ".join(out))
''' % ROOT
for label, env in (("default (F(2x4,3x3) on the fp32 MFMA)", {}),
                   ("IMX_CONV=direct", {"IMX_CONV": "direct"})):
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
    print(label)
    print("  " + "\n  ".join(l for l in r.stdout.strip().splitlines()) if r.returncode == 0 else r.stderr[-800:])

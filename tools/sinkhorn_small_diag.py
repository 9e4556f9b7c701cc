"""Diagnostic (GPU box): VERDICT r3 task 7 -- where do the Sinkhorn-alone errors of small / ragged problems sit?  For each shape: the
library's Z (from its own scores_in, u, v) against a float64 optimal transport on the same scores, and how much of the error is a
row constant plus a column constant (an error of the potentials u, v); tools/sinkhorn_growth_diag.py shows its growth with the
iteration count.  Result (profiles/r04_sinkhorn_*.txt): all of it but ~3e-6."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import util
from image_matching_amd import _lib as L
from image_matching_amd.engine import Engine

shapes = [(7, 64), (39, 32), (71, 43), (15, 32), (1023, 1024), (200, 200)]
d = 128
sd = util.sg_sd(d)
for n0, n1 in shapes:
    eng = Engine(util.sp_config(d, 1024), util.sg_config(d), "cuda")
    eng.load_state_dict(L.NET_SUPERGLUE, sd)
    eng.set_debug(True)
    g = torch.Generator().manual_seed(n0 * 7919 + n1)
    t = {"keypoints0": torch.rand(1, n0, 2, generator=g) * 600, "keypoints1": torch.rand(1, n1, 2, generator=g) * 600,
         "scores0": torch.rand(1, n0, generator=g), "scores1": torch.rand(1, n1, generator=g),
         "descriptors0": torch.nn.functional.normalize(torch.randn(1, d, n0, generator=g), dim=1),
         "descriptors1": torch.nn.functional.normalize(torch.randn(1, d, n1, generator=g), dim=1)}
    eng.superglue(t["keypoints0"], t["scores0"], t["descriptors0"], (1, 1, 480, 640), t["keypoints1"], t["scores1"], t["descriptors1"], (1, 1, 480, 640))
    torch.cuda.synchronize()
    S = eng.fetch("scores_in")[0, :n0, :n1]
    u, v = eng.fetch("u")[0], eng.fetch("v")[0]
    Z = util.transport_Z(S, u, v, n0, n1, float(sd["bin_score"]))
    Zrs, Z64 = util.sinkhorn_fp32_evaluations(S, sd["bin_score"], 30)
    eh = np.abs(Z - Z64)
    print(f"{n0}x{n1}: |u|max {np.abs(u[:n0+1]).max():.0f} |v|max {np.abs(v[:n1+1]).max():.0f} drift bound {util.sinkhorn_drift_bound(u[:n0+1], v[:n1+1], 30):.2e}; "
          f"library err max {eh.max():.2e} rms {np.sqrt((eh**2).mean()):.2e}; oracle fp32 x5: max {max(np.abs(r - Z64).max() for r in Zrs):.2e} rms {max(np.sqrt(((r - Z64)**2).mean()) for r in Zrs):.2e}")
    # row / column structure of the error: how much of it is a per-row constant (u) or a per-column constant (v)?
    e = (Z - Z64)
    print(f"    error decomposition: total rms {np.sqrt((e**2).mean()):.2e}; after removing row means {np.sqrt(((e - e.mean(1, keepdims=True))**2).mean()):.2e}; "
          f"after removing column means {np.sqrt(((e - e.mean(0, keepdims=True))**2).mean()):.2e}; after both {np.sqrt(((e - e.mean(1, keepdims=True) - e.mean(0, keepdims=True) + e.mean())**2).mean()):.2e}")

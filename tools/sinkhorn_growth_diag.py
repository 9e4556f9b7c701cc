"""Diagnostic (GPU box), VERDICT r3 task 7: does the Sinkhorn-alone error of the small / ragged problems GROW with the iteration count
(the drift of marginally stable block modes of the fp32 iteration) or is it there after the first iteration (which would point at the
slab path's partial log-sum-exp merge)?  Same score matrix, sinkhorn_iterations = 1 .. 100, library vs float64, oracle fp32 vs float64;
the error is also split into its additive part e_u(i) + e_v(j) and the rest."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import util
from image_matching_amd import _lib as L
from image_matching_amd.engine import Engine
from oracle import superglue_ref

d = 128
sd = util.sg_sd(d)
for n0, n1 in [(7, 64), (39, 32), (200, 200)]:
    g = torch.Generator().manual_seed(n0 * 7919 + n1)
    t = {"keypoints0": torch.rand(1, n0, 2, generator=g) * 600, "keypoints1": torch.rand(1, n1, 2, generator=g) * 600,
         "scores0": torch.rand(1, n0, generator=g), "scores1": torch.rand(1, n1, generator=g),
         "descriptors0": torch.nn.functional.normalize(torch.randn(1, d, n0, generator=g), dim=1),
         "descriptors1": torch.nn.functional.normalize(torch.randn(1, d, n1, generator=g), dim=1)}
    for iters in (1, 2, 5, 10, 20, 30, 60, 100):
        eng = Engine(util.sp_config(d, 1024), util.sg_config(d, sinkhorn_iterations=iters), "cuda")
        eng.load_state_dict(L.NET_SUPERGLUE, sd)
        eng.set_debug(True)
        eng.superglue(t["keypoints0"], t["scores0"], t["descriptors0"], (1, 1, 480, 640), t["keypoints1"], t["scores1"], t["descriptors1"], (1, 1, 480, 640))
        torch.cuda.synchronize()
        S = eng.fetch("scores_in")[0, :n0, :n1]
        u, v = eng.fetch("u")[0], eng.fetch("v")[0]
        Z = util.transport_Z(S, u, v, n0, n1, float(sd["bin_score"]))
        St = torch.as_tensor(np.ascontiguousarray(S))[None]
        Z64 = superglue_ref.log_optimal_transport(St.double(), sd["bin_score"].double(), iters=iters)[0].numpy()
        Z32 = superglue_ref.log_optimal_transport(St, sd["bin_score"].float(), iters=iters)[0].numpy()
        rms = lambda e: float(np.sqrt((e ** 2).mean()))
        add = lambda e: e - (e - e.mean(1, keepdims=True) - e.mean(0, keepdims=True) + e.mean())
        eh, er = Z - Z64, Z32 - Z64
        print(f"{n0}x{n1} iters {iters:3d}: library rms {rms(eh):.2e} max {np.abs(eh).max():.2e} (non-additive part {rms(eh - add(eh)):.1e}) | "
              f"oracle fp32 rms {rms(er):.2e} max {np.abs(er).max():.2e} (non-additive {rms(er - add(er)):.1e}) | spacing(|u|max={np.abs(u[:n0+1]).max():.0f}) {np.spacing(np.float32(np.abs(u[:n0+1]).max())):.1e}")

#!/usr/bin/env python
"""Where does the library's Sinkhorn differ from a float64 evaluation on its own score matrix?  Prints the potentials' errors per row / column."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from tests import util
from image_matching_amd import _lib as L
from image_matching_amd.engine import Engine
from oracle import superglue_ref

def f64_potentials(S, alpha, iters):
    S = torch.from_numpy(S).double()[None]
    b, m, n = S.shape
    one = S.new_tensor(1)
    ms, ns = (m * one), (n * one)
    a = torch.tensor(alpha, dtype=torch.float64)
    Z = torch.cat([torch.cat([S, a.expand(b, m, 1)], -1), torch.cat([a.expand(b, 1, n), a.expand(b, 1, 1)], -1)], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])[None]
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])[None]
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return u[0].numpy(), v[0].numpy()

d = 128
for n0, n1, iters in ((7, 64, 1), (7, 64, 2), (7, 64, 20)):
    eng = Engine(util.sp_config(d, 1024), util.sg_config(d, sinkhorn_iterations=iters), "cuda")
    sd = util.sg_sd(d)
    eng.load_state_dict(L.NET_SUPERGLUE, sd)
    eng.set_debug(True)
    g = torch.Generator().manual_seed(n0 * 7919 + n1)
    t = {"keypoints0": torch.rand(1, n0, 2, generator=g) * 600, "keypoints1": torch.rand(1, n1, 2, generator=g) * 600,
         "scores0": torch.rand(1, n0, generator=g), "scores1": torch.rand(1, n1, generator=g),
         "descriptors0": torch.nn.functional.normalize(torch.randn(1, d, n0, generator=g), dim=1),
         "descriptors1": torch.nn.functional.normalize(torch.randn(1, d, n1, generator=g), dim=1)}
    t = {k: v.cuda() for k, v in t.items()}
    eng.superglue(t["keypoints0"], t["scores0"], t["descriptors0"], (1, 1, 480, 640), t["keypoints1"], t["scores1"], t["descriptors1"], (1, 1, 480, 640))
    torch.cuda.synchronize()
    S = eng.fetch("scores_in")[0, :n0, :n1]
    u, v = eng.fetch("u")[0][:n0 + 1], eng.fetch("v")[0][:n1 + 1]
    u64, v64 = f64_potentials(S, float(sd["bin_score"]), iters)
    u32, v32 = f64_potentials.__wrapped__(S, float(sd["bin_score"]), iters) if hasattr(f64_potentials, "__wrapped__") else (None, None)
    St = torch.from_numpy(S.copy())[None]
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez(f"gpurun_out/sinkhorn_dbg_{n0}x{n1}_{iters}.npz", S=S, u=u, v=v, alpha=float(sd["bin_score"]))
    print(f"== {n0}x{n1}, {iters} iterations: max|S| {np.abs(S).max():.1f}")
    np.set_printoptions(precision=2, linewidth=200)
    print(" u err x1e4:", (u - u64) * 1e4)
    print(" v err x1e4 (first 16, last):", ((v - v64) * 1e4)[:16], ((v - v64) * 1e4)[-1])
    print(" u64:", u64[:10], " v64[:6]:", v64[:6])

"""Diagnostic (GPU box): where does the images-in path leave the 1e-4 + 1e-4|ref| bar on the strict weight set?  For the first pairs
of a strict fixture: the HIP SuperPoint outputs against the oracle's, then the library's gnn17 / scores_in / Z (taps of one
imx_match_pairs call) against the oracle's SuperGlue on (a) the reference's inputs and (b) the library's OWN SuperPoint outputs."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import util
from tests.test_gpu_strict import _matching_t
from oracle import superglue_ref, superpoint_ref

name, B = (sys.argv[1] if len(sys.argv) > 1 else "strict_c3.npz"), int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = util.golden(name)
H, W, d, K = (int(g[k]) for k in ("H", "W", "d", "K"))
m = _matching_t(d, K)
ims = [util.pair(int(seed), H, W) for seed in g["seeds"][:B]]
i0, i1 = torch.cat([p[0] for p in ims]).cuda(), torch.cat([p[1] for p in ims]).cuda()
eng = m._shared.get_engine([0, 1])
eng.set_debug(True)
out = m.match_batch(i0, i1, want_desc=True)
torch.cuda.synchronize()
X, S, U, V = eng.fetch("x"), eng.fetch("scores_in"), eng.fetch("u"), eng.fetch("v")
sd_sp, sd_sg = util.sp_sd(d), util.sg_sd(d, variant="t")
alpha = float(sd_sg["bin_score"])
Kp = (K + 31) // 32 * 32
def used(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    r = np.abs(a - b) / (1e-4 + 1e-4 * np.abs(b))
    i = np.unravel_index(r.argmax(), r.shape)
    return f"used {r.max():.3f} (outside {int((r > 1).sum())}/{r.size}; at {i}: |ref| {abs(b[i]):.3g} err {abs(a[i]-b[i]):.2e})"
for b in range(B):
    seed = int(g["seeds"][b])
    x0, x1 = ims[b]
    data_ref, data_own = {}, {}
    for side, x in (("0", x0), ("1", x1)):
        o = superpoint_ref.superpoint_forward(x, sd_sp, util.sp_config(d, K), return_dense=True)
        kp_ref = o["keypoints"][0]
        kp = out["keypoints" + side][b].cpu()
        sc = out["scores" + side][b].cpu()
        ds = out["descriptors" + side][b].cpu().t()
        same_order = bool((kp == kp_ref).all())
        print(f"seed {seed} side {side}: keypoint order identical {same_order}; scores err max {float((sc - o['scores'][0]).abs().max()) if same_order else float('nan'):.2e}; "
              f"descriptors err max {float((ds - o['descriptors'][0]).abs().max()) if same_order else float('nan'):.2e}")
        data_ref["keypoints" + side], data_ref["scores" + side], data_ref["descriptors" + side] = kp_ref[None], o["scores"][0][None], o["descriptors"][0][None]
        data_own["keypoints" + side], data_own["scores" + side], data_own["descriptors" + side] = kp[None], sc[None], ds[None]
    for d_ in (data_ref, data_own):
        d_["image_shape0"] = d_["image_shape1"] = (1, 1, H, W)
    g0 = X[b * Kp:b * Kp + K].T
    g1 = X[B * Kp + b * Kp:B * Kp + b * Kp + K].T
    Z = util.transport_Z(S[b], U[b], V[b], K, K, alpha)
    for label, dat in (("oracle SG on the oracle's SP outputs", data_ref), ("oracle SG on the library's OWN SP outputs", data_own)):
        dn = superglue_ref.superglue_forward(dat, sd_sg, util.sg_config(d), return_dense=True)["dense"]
        print(f"  vs {label}: gnn17 {used(np.stack([g0, g1]), np.stack([dn['gnn0'][0].numpy(), dn['gnn1'][0].numpy()]))}")
        print(f"     scores_in {used(S[b, :K, :K], dn['scores_in'][0].numpy())}")
        print(f"     Z {used(Z, dn['Z'][0].numpy())}")

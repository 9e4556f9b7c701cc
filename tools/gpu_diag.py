#!/usr/bin/env python
"""Stage-by-stage parity report (HIP path vs oracle / goldens) — prints error statistics instead
of asserting, for bring-up on the GPU box.  Test infrastructure: imports oracle/."""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from image_matching_amd import _lib as L                                     # noqa: E402
from image_matching_amd.engine import Engine                                 # noqa: E402
from oracle import superglue_ref, superpoint_ref                             # noqa: E402
from tests import util                                                       # noqa: E402


def stat(name, a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        print(f"  {name:28s} SHAPE {a.shape} vs {b.shape}")
        return
    err = np.abs(a - b)
    tol = 1e-4 + 1e-4 * np.abs(b)
    print(f"  {name:28s} max|ref| {np.abs(b).max():10.4g}  max err {err.max():10.3e}  mean err {err.mean():10.3e}  "
          f"out-of-tol {int((err > tol).sum())}/{err.size}  nan {int(np.isnan(a).sum())}", flush=True)


def nhwc_to_nchw(a):
    return np.transpose(a, (0, 3, 1, 2))


def diag_superpoint(name, H, W, seed, K, d=128, B2=True):
    print(f"== SuperPoint {name}: {H}x{W} seed {seed} K {K} d {d}")
    x0, x1 = util.pair(seed, H, W)
    x = torch.cat([x0, x1]) if B2 else x0
    sd = util.sp_sd(d)
    ref = superpoint_ref.superpoint_forward(x, sd, util.sp_config(d, K), return_dense=True)
    eng = Engine(util.sp_config(d, K), util.sg_config(d), "cuda")
    eng.load_state_dict(L.NET_SUPERPOINT, sd)
    t = time.time()
    kpts, scores, desc, n = eng.superpoint(x.cuda())
    torch.cuda.synchronize()
    print(f"  forward ok in {time.time() - t:.3f}s; counts {n} (ref {[len(s) for s in ref['scores']]})")
    # encoder taps against torch (oracle pieces)
    with torch.no_grad():
        import torch.nn.functional as F
        x1_ = superpoint_ref._double_conv(x, sd, "inc.conv.conv")
        x2_ = superpoint_ref._double_conv(F.max_pool2d(x1_, 2), sd, "down1.mpconv.1.conv")
        x3_ = superpoint_ref._double_conv(F.max_pool2d(x2_, 2), sd, "down2.mpconv.1.conv")
    stat("a1 = pool(x1)", nhwc_to_nchw(eng.fetch("a1")), F.max_pool2d(x1_, 2).numpy())
    stat("a2 = pool(x2)", nhwc_to_nchw(eng.fetch("a2")), F.max_pool2d(x2_, 2).numpy())
    stat("a3 = pool(x3)", nhwc_to_nchw(eng.fetch("a3")), F.max_pool2d(x3_, 2).numpy())
    stat("x4", nhwc_to_nchw(eng.fetch("x4")), ref["x4"].numpy())
    stat("semi", nhwc_to_nchw(eng.fetch("semi")), ref["semi"].numpy())
    raw = nhwc_to_nchw(eng.fetch("desc_raw"))
    stat("desc (normalised)", raw / np.linalg.norm(raw, axis=1, keepdims=True), ref["desc"].numpy())
    stat("score_map", eng.fetch("score_map"), ref["score_map"].numpy())
    my_nms = eng.fetch("nms")
    stat("nms (own map)", my_nms, ref["nms"].numpy())
    nms_ref_in = eng.op_nms(ref["score_map"], 4).cpu().numpy()
    print(f"  nms on oracle map bit-exact: {np.array_equal(nms_ref_in, ref['nms'].numpy())} "
          f"(mismatches {int((nms_ref_in != ref['nms'].numpy()).sum())})")
    for b in range(x.shape[0]):
        kr, sr, dr = ref["keypoints"][b].numpy(), ref["scores"][b].numpy(), ref["descriptors"][b].numpy()
        km, sm, dm = kpts[b, :n[b]].cpu().numpy(), scores[b, :n[b]].cpu().numpy(), desc[b, :n[b]].t().cpu().numpy()
        same_order = km.shape == kr.shape and np.array_equal(km, kr)
        set_r = set(map(tuple, kr.astype(int)))
        set_m = set(map(tuple, km.astype(int)))
        print(f"  img{b}: kpts same order {same_order}; set equal {set_r == set_m} (only-ref {len(set_r - set_m)}, only-mine {len(set_m - set_r)})")
        if set_r == set_m:
            a = util.canon_keypoints(km, sm, dm)
            r = util.canon_keypoints(kr, sr, dr)
            stat(f"img{b} scores (canon)", a[1], r[1])
            stat(f"img{b} descriptors (canon)", a[2], r[2])
    return eng


def diag_superglue(name, gname):
    print(f"== SuperGlue {name} ({gname})")
    g = util.golden(gname)
    sd = util.sg_sd(128)
    eng = Engine(util.sp_config(128, 1024), util.sg_config(128), "cuda")
    eng.load_state_dict(L.NET_SUPERGLUE, sd)
    eng.set_debug(True)
    t = {k: torch.from_numpy(g[k]).cuda() for k in ("keypoints0", "keypoints1", "scores0", "scores1", "descriptors0", "descriptors1")}
    shp = (1, 1, 120, 160)
    m0, m1, ms0, ms1 = eng.superglue(t["keypoints0"], t["scores0"], t["descriptors0"], shp,
                                     t["keypoints1"], t["scores1"], t["descriptors1"], shp)
    torch.cuda.synchronize()
    N0, N1 = g["keypoints0"].shape[1], g["keypoints1"].shape[1]
    N0p, N1p = (N0 + 31) // 32 * 32, (N1 + 31) // 32 * 32

    def rows(a):   # internal rows -> (side0 (d,N0), side1 (d,N1))
        return a[:N0].T, a[N0p:N0p + N1].T
    k0, k1 = rows(eng.fetch("kenc"))
    stat("kenc0", k0, g["kenc0"][0]); stat("kenc1", k1, g["kenc1"][0])
    a0, a1 = rows(eng.fetch("gnn0"))
    stat("gnn layer0 side0", a0, g["tap0_0"][0]); stat("gnn layer0 side1", a1, g["tap0_1"][0])
    a0, a1 = rows(eng.fetch("gnn1"))
    stat("gnn layer1 side0", a0, g["tap1_0"][0]); stat("gnn layer1 side1", a1, g["tap1_1"][0])
    a0, a1 = rows(eng.fetch("gnn17"))
    stat("gnn out side0", a0, g["gnn0"][0]); stat("gnn out side1", a1, g["gnn1"][0])
    S = eng.fetch("scores_in")[0, :N0, :N1]
    stat("scores_in", S, g["scores_in"][0])
    u, v = eng.fetch("u")[0], eng.fetch("v")[0]
    norm = -np.log(np.float32(N0 + N1))
    Z = np.full((N0 + 1, N1 + 1), float(sd["bin_score"]), dtype=np.float32)
    Z[:N0, :N1] = S
    Z = (Z + u[:N0 + 1, None]) + v[None, :N1 + 1] - norm
    stat("Z", Z, g["Z"][0])
    print(f"  matches0 equal {np.array_equal(m0.cpu().numpy(), g['matches0'])} "
          f"(diff {int((m0.cpu().numpy() != g['matches0']).sum())}, matched ref {(g['matches0'] > -1).sum()})  "
          f"matches1 equal {np.array_equal(m1.cpu().numpy(), g['matches1'])}")
    stat("matching_scores0", ms0.cpu().numpy(), g["matching_scores0"])
    stat("matching_scores1", ms1.cpu().numpy(), g["matching_scores1"])


def main():
    print(L.load_library().imx_version().decode(), torch.cuda.get_device_name(0))
    steps = [
        lambda: diag_superpoint("small", 120, 160, 12, 207),
        lambda: diag_superpoint("ragged", 123, 165, 2, -1),
        lambda: diag_superglue("small", "sg_small.npz"),
        lambda: diag_superpoint("C3", 480, 640, 59, 1024),
    ]
    for s in steps:
        try:
            s()
        except Exception:
            traceback.print_exc()
        sys.stdout.flush()


if __name__ == "__main__":
    main()

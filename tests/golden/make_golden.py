#!/usr/bin/env python
"""Generate golden vectors by IMPORTING THE REFERENCE (build container only).

Run:  python tests/golden/make_golden.py            (needs /root/reference; never runs on the GPU box)

Writes
  image-matching_amd/data/synth_bn_stats.npz   calibrated BatchNorm running stats for the
                                               synthetic weight sets (synth.py seeds below)
  tests/golden/*.npz                           inputs are re-derivable from seeds (synth.py is
                                               portable); files hold the reference's OUTPUTS.

The reference modules are imported unchanged; only `torch.load` is bypassed for the official
SuperPoint (its weights file is a git-LFS pointer).  Seeds whose discrete decisions sit close
to a tie (top-k boundary, match argmax margin, match threshold) are rejected — see MARGIN_*.
"""
import os
import sys
import warnings

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)
warnings.filterwarnings("ignore")

from superpoint.models.superpoint_test import SuperPoint  # noqa: E402  (reference)
from superpoint.models import superpoint_test as ref_sp_mod  # noqa: E402
from superglue.models.superglue_test import SuperGlue  # noqa: E402  (reference)
from superglue.models.matching_test import Matching  # noqa: E402  (reference)
from image_matching_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)

SP_SEED, SG_SEED, SG_GAINS = synth.SP_SEED, synth.SG_SEED, synth.SG_GAINS
MARGIN_MATCH = 5e-3     # small cases: min decision-relevant (top1 - top2) gap in Z, and |mscore - thr|
                        # (fp32-vs-fp64 noise of the reference itself on Z is ~5e-4 at N=1024)
MARGIN_TOPK = 1e-4      # min gap between kept/dropped scores at the top-k boundary
OUT = os.path.join(REPO, "tests", "golden")
DATA = os.path.join(REPO, "image-matching_amd", "data")


def to_torch(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def calibrate(model, fwd):
    """One train()-mode forward with momentum=1.0: running stats := batch stats."""
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.momentum = 1.0
    model.train()
    fwd()
    model.eval()
    return {k: v.numpy().copy() for k, v in model.state_dict().items()
            if k.endswith("running_mean") or k.endswith("running_var")}


def build_sp(d, max_kp, stats=None, **kw):
    sp = SuperPoint({"weights": None, "descriptor_dim": d, "max_keypoints": max_kp, **kw})
    sd = synth.synth_state_dict(synth.superpoint_bn_shapes(d), SP_SEED)
    if stats is not None:
        synth.apply_bn_stats(sd, stats)
    sp.load_state_dict(to_torch(sd))
    return sp.eval(), sd


def build_sg(d, kenc, iters, thr, stats=None, bin_score=None, layers=None, gains=None):
    cfg = {"weights": None, "descriptor_dim": d, "keypoint_encoder": list(kenc),
           "sinkhorn_iterations": iters, "match_threshold": thr}
    if layers is not None:
        cfg["GNN_layers"] = layers
    sg = SuperGlue(cfg)
    nl = len(sg.config["GNN_layers"])
    sd = synth.synth_state_dict(synth.superglue_shapes(d, kenc, nl), SG_SEED, gains=SG_GAINS if gains is None else gains)
    if stats is not None:
        synth.apply_bn_stats(sd, stats)
    if bin_score is not None:
        sd["bin_score"] = np.float32(bin_score).reshape(())
    sg.load_state_dict(to_torch(sd))
    return sg.eval(), sd


def pair_tensor(seed, H, W):
    im0, im1 = synth.synth_pair(seed, H, W)
    return torch.from_numpy(im0)[None, None], torch.from_numpy(im1)[None, None]


def sg_data(x0, x1, o0, o1):
    return {"image0": x0, "image1": x1,
            "keypoints0": o0["keypoints"][0][None], "keypoints1": o1["keypoints"][0][None],
            "scores0": o0["scores"][0][None], "scores1": o1["scores"][0][None],
            "descriptors0": o0["descriptors"][0][None], "descriptors1": o1["descriptors"][0][None]}


def sg_dense(sg, data):
    """Re-run the reference's own sub-modules to capture intermediates (same calls as forward)."""
    from superglue.models.superglue_test import normalize_keypoints, log_optimal_transport
    k0 = normalize_keypoints(data["keypoints0"], data["image0"].shape)
    k1 = normalize_keypoints(data["keypoints1"], data["image1"].shape)
    d0 = data["descriptors0"] + sg.kenc(k0, data["scores0"])
    d1 = data["descriptors1"] + sg.kenc(k1, data["scores1"])
    kenc0, kenc1 = d0.clone(), d1.clone()
    taps = {}
    for i, (layer, name) in enumerate(zip(sg.gnn.layers, sg.gnn.names)):
        s0, s1 = (d1, d0) if name == "cross" else (d0, d1)
        delta0, delta1 = layer(d0, s0), layer(d1, s1)
        d0, d1 = d0 + delta0, d1 + delta1
        taps[i] = (d0.clone(), d1.clone())
    m0, m1 = sg.final_proj(d0), sg.final_proj(d1)
    sc = torch.einsum("bdn,bdm->bnm", m0, m1) / sg.config["descriptor_dim"] ** .5
    Z = log_optimal_transport(sc, sg.bin_score, iters=sg.config["sinkhorn_iterations"])
    return {"kenc0": kenc0, "kenc1": kenc1, "taps": taps, "gnn0": d0, "gnn1": d1,
            "mdesc0": m0, "mdesc1": m1, "scores_in": sc, "Z": Z}


def sg_dense_f64(sg, data):
    """The same reference sub-module calls evaluated in float64 (a deep copy of the module cast with .double()):
    the anchor the fp32 implementations (reference fp32, HIP fp32) are both measured against."""
    import copy
    sg64 = copy.deepcopy(sg).double()
    d64 = {k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in data.items()}
    out = sg_dense(sg64, d64)
    return {k: v for k, v in out.items() if k != "taps"}


def envelope(a32, a64):
    """(max, rms) of |fp32 - fp64| -- the reference's own fp32 rounding envelope on a tensor."""
    e = (a32.double() - a64).abs()
    return np.array([float(e.max()), float(e.pow(2).mean().sqrt())])


def match_margins(Z, r, thr):
    """min top1-top2 gap over rows/cols that produced a match, min |mscore-thr| over mutual."""
    Zi = Z[0, :-1, :-1]
    t0 = Zi.topk(2, dim=1).values
    t1 = Zi.topk(2, dim=0).values
    m0, m1 = r["matches0"][0], r["matches1"][0]
    ms0 = r["matching_scores0"][0]
    # every row/col argmax participates in the mutual test, so use ALL rows/cols for argmax margin
    g0 = (t0[:, 0] - t0[:, 1]).min().item()
    g1 = (t1[0] - t1[1]).min().item()
    gm0 = (t0[:, 0] - t0[:, 1])[m0 > -1].min().item() if (m0 > -1).any() else float("inf")
    mut = ms0 > 0
    gthr = (ms0[mut] - thr).abs().min().item() if mut.any() else float("inf")
    # decision-relevant gap: a row's (col's) argmax tie matters only if its top-1 is currently
    # mutual, or if flipping to its top-2 would create a mutual pair.
    i0 = Zi.topk(2, dim=1).indices            # (M,2)
    i1 = Zi.topk(2, dim=0).indices            # (2,N)
    am0, am1 = i0[:, 0], i1[0]
    rows = torch.arange(Zi.shape[0]); cols = torch.arange(Zi.shape[1])
    rel0 = (am1[i0[:, 0]] == rows) | (am1[i0[:, 1]] == rows)
    rel1 = (am0[i1[0]] == cols) | (am0[i1[1]] == cols)
    gd = min((t0[:, 0] - t0[:, 1])[rel0].min().item() if rel0.any() else float("inf"),
             (t1[0] - t1[1])[rel1].min().item() if rel1.any() else float("inf"))
    return {"argmax_gap_all": min(g0, g1), "argmax_gap_matched": gm0, "thr_gap": gthr,
            "decision_gap": gd}


def topk_margin(sp_out_all_scores, k):
    s = np.sort(sp_out_all_scores)[::-1]
    if len(s) <= k:
        return float("inf"), 0.0
    gaps = s[:k][:-1] - s[:k][1:]
    return float(s[k - 1] - s[k]), float(gaps.min())


def sp_dense(sp, x):
    """Re-run the reference module's layers to capture dense intermediates (same ops as forward)."""
    x1 = sp.inc(x); x2 = sp.down1(x1); x3 = sp.down2(x2); x4 = sp.down3(x3)
    cPa = sp.relu(sp.bnPa(sp.convPa(x4))); semi = sp.bnPb(sp.convPb(cPa))
    cDa = sp.relu(sp.bnDa(sp.convDa(x4))); desc = sp.bnDb(sp.convDb(cDa))
    dn = torch.norm(desc, p=2, dim=1); desc = desc.div(torch.unsqueeze(dn, 1))
    scores = torch.nn.functional.softmax(semi, 1)[:, :-1]
    b, _, h, w = scores.shape
    scores = scores.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
    nms = ref_sp_mod.simple_nms(scores, sp.config["nms_radius"])
    return {"x1": x1, "x2": x2, "x3": x3, "x4": x4, "semi": semi, "desc": desc, "score_map": scores, "nms": nms}


def npz(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()})
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.0f} KB")


def main():
    os.makedirs(DATA, exist_ok=True)
    stats_all = {}

    # ---------------------------------------------------------------- calibration (C3 shape)
    print("calibrating BN stats (480x640, pair seed 0) ...")
    x0, x1 = pair_tensor(0, 480, 640)
    xb = torch.cat([x0, x1])
    for d in (128, 256):
        sp, _ = build_sp(d, 1024 if d == 128 else 2048)
        st = calibrate(sp, lambda: sp(xb))
        for k, v in st.items():
            stats_all[f"sp{d}/{k}"] = v
    sp_stats = {d: {k.split("/", 1)[1]: v for k, v in stats_all.items() if k.startswith(f"sp{d}/")} for d in (128, 256)}

    sg_cfgs = {128: ([32, 64, 128], 30, 0.1), 256: ([32, 64, 128, 256], 100, 0.2)}
    sg_stats, sg_bin = {}, {}
    for d, (kenc, iters, thr) in sg_cfgs.items():
        sp, _ = build_sp(d, 1024, sp_stats[d])
        o0, o1 = sp(x0), sp(x1)
        data = sg_data(x0, x1, o0, o1)
        sg, _ = build_sg(d, kenc, iters, thr)
        st = calibrate(sg, lambda: sg(data))
        sg_stats[d] = st
        for k, v in st.items():
            stats_all[f"sg{d}/{k}"] = v
        dn = sg_dense(sg, data)
        sc = dn["scores_in"]
        sg_bin[d] = float(sc.mean() + 2.0 * sc.std())                      # mean+2σ
        stats_all[f"sg{d}/bin_score"] = np.float32(sg_bin[d])
        print(f"  d={d}: scores_in mean {sc.mean():.3f} std {sc.std():.3f} max {sc.max():.2f} -> bin_score {sg_bin[d]:.4f}")
    # descriptor_dim 64 / keypoint_encoder [32, 64] (reference README.md:134-140), added in round 2 AFTER the 128/256
    # entries so those stay bit-identical
    sg_cfgs[64] = (synth.SG_CONFIGS[64][0], synth.SG_CONFIGS[64][1], synth.SG_CONFIGS[64][2])
    sp, _ = build_sp(64, 1024)
    st = calibrate(sp, lambda: sp(xb))
    sp_stats[64] = st
    for k, v in st.items():
        stats_all[f"sp64/{k}"] = v
    sp, _ = build_sp(64, 1024, sp_stats[64])
    o0, o1 = sp(x0), sp(x1)
    data = sg_data(x0, x1, o0, o1)
    sg, _ = build_sg(64, *sg_cfgs[64])
    st = calibrate(sg, lambda: sg(data))
    sg_stats[64] = st
    for k, v in st.items():
        stats_all[f"sg64/{k}"] = v
    sc = sg_dense(sg, data)["scores_in"]
    sg_bin[64] = float(sc.mean() + 2.0 * sc.std())
    stats_all["sg64/bin_score"] = np.float32(sg_bin[64])
    print(f"  d=64: scores_in mean {sc.mean():.3f} std {sc.std():.3f} max {sc.max():.2f} -> bin_score {sg_bin[64]:.4f}")
    np.savez_compressed(os.path.join(DATA, "synth_bn_stats.npz"), **stats_all)
    print("  wrote synth_bn_stats.npz", os.path.getsize(os.path.join(DATA, "synth_bn_stats.npz")) // 1024, "KB")

    # ---------------------------------------------------------------- small SuperPoint, dense
    for name, H, W, K, seed in (("sp_small", 120, 160, 200, 12), ("sp_ragged", 123, 165, -1, 2)):
        print(name)
        sp, _ = build_sp(128, K, sp_stats[128])
        xa, xb_ = pair_tensor(seed, H, W)
        x = torch.cat([xa, xb_])
        dn = sp_dense(sp, x)
        if K >= 0:
            # pick the top-k cut (near the requested K) whose kept/dropped score gap is largest on
            # both images, so the kept SET is robust to fp32 rounding differences
            cands = [np.sort(ref_sp_mod.remove_borders(torch.nonzero(s > sp.config["keypoint_threshold"]),
                                                       s[s > sp.config["keypoint_threshold"]], 4, s.shape[0], s.shape[1])[1].numpy())[::-1]
                     for s in dn["nms"]]
            K = max(range(K - 20, K + 21), key=lambda k: min(c[k - 1] - c[k] for c in cands))
            sp, _ = build_sp(128, K, sp_stats[128])
        o = sp(x)
        allsc = [ref_sp_mod.remove_borders(torch.nonzero(s > sp.config["keypoint_threshold"]),
                                           s[s > sp.config["keypoint_threshold"]], 4, s.shape[0], s.shape[1])[1].numpy()
                 for s in dn["nms"]]
        for b in range(2):
            gb, gmin = topk_margin(allsc[b], K if K >= 0 else 10 ** 9)
            print(f"  img{b}: candidates {len(allsc[b])} kept {len(o['scores'][b])} boundary gap {gb:.3g}")
            assert gb > MARGIN_TOPK, "top-k boundary too close to a tie; pick another seed"
        npz(name + ".npz", H=H, W=W, seed=seed, max_keypoints=K,
            x4=dn["x4"], semi=dn["semi"], desc=dn["desc"], score_map=dn["score_map"], nms=dn["nms"],
            x1_sub=dn["x1"][:, ::8, ::4, ::4], x2_sub=dn["x2"][:, ::4, ::2, ::2], x3=dn["x3"][:, ::2],
            **{f"keypoints{b}": o["keypoints"][b] for b in range(2)},
            **{f"scores{b}": o["scores"][b] for b in range(2)},
            **{f"descriptors{b}": o["descriptors"][b] for b in range(2)})

    # ---------------------------------------------------------------- official (no-BN) SuperPoint
    print("sp_official (superglue/models/superpoint.py, d=256, torch.load bypassed)")
    from superglue.models import superpoint as ref_off_mod
    sd_off = synth.synth_state_dict(synth.superpoint_official_shapes(256), 77)
    real_load = torch.load
    torch.load = lambda *a, **k: to_torch(sd_off)        # weights/superpoint_v1.pth is an LFS pointer
    try:
        spo = ref_off_mod.SuperPoint({"descriptor_dim": 256, "max_keypoints": 300}).eval()
    finally:
        torch.load = real_load
    xo = pair_tensor(5, 128, 192)[0]
    oo = spo({"image": xo})
    npz("sp_official.npz", H=128, W=192, seed=5, max_keypoints=300, weight_seed=77,
        keypoints0=oo["keypoints"][0], scores0=oo["scores"][0], descriptors0=oo["descriptors"][0])

    # ---------------------------------------------------------------- align_corners=True unit
    print("sample_descriptors align_corners=True (torch.__version__ patched to 1.7.1)")
    dmap = torch.from_numpy(synth.normal(7, "dmap", 32 * 15 * 20).reshape(1, 32, 15, 20))
    kp = torch.from_numpy(np.stack([synth.uniform(7, "kx", 300) * 159, synth.uniform(7, "ky", 300) * 119], 1))[None]
    real_v = torch.__version__
    out_false = ref_sp_mod.sample_descriptors(kp.clone(), dmap, 8)
    torch.__version__ = "1.7.1"
    try:
        out_true = ref_sp_mod.sample_descriptors(kp.clone(), dmap, 8)
    finally:
        torch.__version__ = real_v
    npz("sample_desc.npz", dmap=dmap, kp=kp, out_false=out_false, out_true=out_true)

    # ---------------------------------------------------------------- small SuperGlue, dense
    print("sg_small (keypoints from sp_small pair, d=128, 18 layers, 30 iters)")
    sp, _ = build_sp(128, 200, sp_stats[128])
    xa, xb_ = pair_tensor(12, 120, 160)
    o0, o1 = sp(xa), sp(xb_)
    data = sg_data(xa, xb_, o0, o1)
    sg, _ = build_sg(128, [32, 64, 128], 30, 0.1, sg_stats[128], sg_bin[128])
    r = sg(data)
    dn = sg_dense(sg, data)
    mg = match_margins(dn["Z"], r, 0.1)
    print("  matches", int((r["matches0"] > -1).sum()), mg)
    assert mg["decision_gap"] > MARGIN_MATCH and mg["thr_gap"] > MARGIN_MATCH, mg
    npz("sg_small.npz", keypoints0=data["keypoints0"], keypoints1=data["keypoints1"],
        scores0=data["scores0"], scores1=data["scores1"],
        descriptors0=data["descriptors0"], descriptors1=data["descriptors1"],
        kenc0=dn["kenc0"], kenc1=dn["kenc1"], tap0_0=dn["taps"][0][0], tap0_1=dn["taps"][0][1],
        tap1_0=dn["taps"][1][0], tap1_1=dn["taps"][1][1], gnn0=dn["gnn0"], gnn1=dn["gnn1"],
        scores_in=dn["scores_in"], Z=dn["Z"], matches0=r["matches0"], matches1=r["matches1"],
        matching_scores0=r["matching_scores0"], matching_scores1=r["matching_scores1"],
        **{f"margin_{k}": v for k, v in mg.items()})

    # unequal keypoint counts + empty set early-out dtypes
    print("sg_ragged (N0=150, N1=97) and empty-set")
    d2 = {k: (v[:, :150] if k.endswith("0") and k != "image0" and v.dim() == 2 else v) for k, v in data.items()}
    d2 = dict(data)
    d2["keypoints0"], d2["scores0"], d2["descriptors0"] = data["keypoints0"][:, :150], data["scores0"][:, :150], data["descriptors0"][:, :, :150]
    d2["keypoints1"], d2["scores1"], d2["descriptors1"] = data["keypoints1"][:, :97], data["scores1"][:, :97], data["descriptors1"][:, :, :97]
    r2 = sg(d2)
    dn2 = sg_dense(sg, d2)
    mg2 = match_margins(dn2["Z"], r2, 0.1)
    print("  matches", int((r2["matches0"] > -1).sum()), mg2)
    d3 = dict(d2)
    d3["keypoints1"], d3["scores1"], d3["descriptors1"] = data["keypoints1"][:, :0], data["scores1"][:, :0], data["descriptors1"][:, :, :0]
    r3 = sg(d3)
    npz("sg_ragged.npz", n0=150, n1=97, Z=dn2["Z"], matches0=r2["matches0"], matches1=r2["matches1"],
        matching_scores0=r2["matching_scores0"], matching_scores1=r2["matching_scores1"],
        empty_matches0=r3["matches0"], empty_matches1=r3["matches1"],
        empty_scores0=r3["matching_scores0"], empty_scores1=r3["matching_scores1"],
        empty_dtype=str(r3["matches0"].dtype), normal_dtype=str(r2["matches0"].dtype),
        **{f"margin_{k}": v for k, v in mg2.items()})

    anchors = {}          # fp64 anchors (round 2): filled below, written as fp64_anchor.npz
    a64 = sg_dense_f64(sg, data)
    for k in ("gnn0", "gnn1", "scores_in", "Z"):
        anchors[f"sg_small/{k}_f64"] = a64[k]
        anchors[f"sg_small/{k}_ref32_err"] = envelope(dn[k], a64[k])

    # ---------------------------------------------------------------- C3 end to end (Matching)
    for name, H, W, d, K, seeds in (("c3_pair", 480, 640, 128, 1024, (59, 55)),
                                    ("c5_pair", 960, 1280, 256, 2048, (19,))):
        kenc, iters, thr = sg_cfgs[d]
        cfg = {"superpoint": {"weights": None, "descriptor_dim": d, "nms_radius": 4,
                              "keypoint_threshold": 0.005, "max_keypoints": K},
               "superglue": {"weights": None, "descriptor_dim": d, "keypoint_encoder": kenc,
                             "sinkhorn_iterations": iters, "match_threshold": thr}}
        m = Matching(cfg).eval()
        _, sd_sp = build_sp(d, K, sp_stats[d])
        _, sd_sg = build_sg(d, kenc, iters, thr, sg_stats[d], sg_bin[d])
        m.superpoint.load_state_dict(to_torch(sd_sp))
        m.superglue.load_state_dict(to_torch(sd_sg))
        for seed in seeds:
            print(f"{name} seed {seed}")
            xa, xb_ = pair_tensor(seed, H, W)
            pred = m({"image0": xa, "image1": xb_})
            data = {"image0": xa, "image1": xb_, **{k: torch.stack(list(v)) for k, v in pred.items() if isinstance(v, (list, tuple))}}
            dn = sg_dense(m.superglue, data)
            mg = match_margins(dn["Z"], pred, thr)
            nm = int((pred["matches0"] > -1).sum())
            print(f"  kpts {len(pred['scores0'][0])}/{len(pred['scores1'][0])} matches {nm} {mg}")
            assert len(pred["scores0"][0]) == K and len(pred["scores1"][0]) == K
            # large N: the minimum over ~N rows is statistically small; seeds were searched for the best margins
            assert mg["decision_gap"] > 5e-4 and mg["thr_gap"] > 5e-4, mg
            sub = slice(0, None, 16)
            npz(f"{name}_s{seed}.npz", H=H, W=W, d=d, K=K, seed=seed,
                keypoints0=pred["keypoints0"][0], keypoints1=pred["keypoints1"][0],
                scores0=pred["scores0"][0], scores1=pred["scores1"][0],
                descriptors0_sub=pred["descriptors0"][0][:, sub], descriptors1_sub=pred["descriptors1"][0][:, sub],
                matches0=pred["matches0"], matches1=pred["matches1"],
                matching_scores0=pred["matching_scores0"], matching_scores1=pred["matching_scores1"],
                Z_sub=dn["Z"][0, ::8, ::8], scores_in_sub=dn["scores_in"][0, ::8, ::8],
                out_dtypes=str({k: (str(v.dtype) if isinstance(v, torch.Tensor) else type(v).__name__ + ":" + str(v[0].dtype)) for k, v in pred.items()}),
                **{f"margin_{k}": v for k, v in mg.items()})
            # fp64 anchor of the same forward (reference module in double, same keypoints/descriptors)
            a64 = sg_dense_f64(m.superglue, data)
            tag = f"{name}_s{seed}"
            anchors[f"{tag}/scores_in_sub_f64"] = a64["scores_in"][0, ::8, ::8]
            anchors[f"{tag}/Z_sub_f64"] = a64["Z"][0, ::8, ::8]
            anchors[f"{tag}/scores_in_ref32_err"] = envelope(dn["scores_in"][0, ::8, ::8], a64["scores_in"][0, ::8, ::8])
            anchors[f"{tag}/Z_ref32_err"] = envelope(dn["Z"][0, ::8, ::8], a64["Z"][0, ::8, ::8])
            print("  fp32-vs-fp64 (max, rms): scores_in", anchors[f"{tag}/scores_in_ref32_err"], "Z", anchors[f"{tag}/Z_ref32_err"])
    npz("fp64_anchor.npz", **anchors)

    # ---------------------------------------------------------------- descriptor_dim 64 (HD = 16), small, dense
    print("sg_small_d64 (d=64, keypoint_encoder [32,64], reference README.md:134-140)")
    kenc, iters, thr = sg_cfgs[64]
    sp, _ = build_sp(64, 200, sp_stats[64])
    xa, xb_ = pair_tensor(12, 120, 160)
    dn_sp = sp_dense(sp, torch.cat([xa, xb_]))
    cands = [np.sort(ref_sp_mod.remove_borders(torch.nonzero(s_ > 0.005), s_[s_ > 0.005], 4, s_.shape[0], s_.shape[1])[1].numpy())[::-1]
             for s_ in dn_sp["nms"]]
    K64 = max(range(180, 221), key=lambda k: min(c[k - 1] - c[k] for c in cands))
    sp, _ = build_sp(64, K64, sp_stats[64])
    o0, o1 = sp(xa), sp(xb_)
    data = sg_data(xa, xb_, o0, o1)
    sg, _ = build_sg(64, kenc, iters, thr, sg_stats[64], sg_bin[64])
    r = sg(data)
    dn = sg_dense(sg, data)
    mg = match_margins(dn["Z"], r, thr)
    print("  K", K64, "matches", int((r["matches0"] > -1).sum()), mg)
    a64 = sg_dense_f64(sg, data)
    npz("sg_small_d64.npz", H=120, W=160, seed=12, max_keypoints=K64,
        keypoints0=data["keypoints0"], keypoints1=data["keypoints1"], scores0=data["scores0"], scores1=data["scores1"],
        descriptors0=data["descriptors0"], descriptors1=data["descriptors1"],
        kenc0=dn["kenc0"], kenc1=dn["kenc1"], tap0_0=dn["taps"][0][0], tap0_1=dn["taps"][0][1],
        gnn0=dn["gnn0"], gnn1=dn["gnn1"], scores_in=dn["scores_in"], Z=dn["Z"],
        gnn0_f64=a64["gnn0"], gnn1_f64=a64["gnn1"], scores_in_f64=a64["scores_in"], Z_f64=a64["Z"],
        matches0=r["matches0"], matches1=r["matches1"],
        matching_scores0=r["matching_scores0"], matching_scores1=r["matching_scores1"],
        **{f"margin_{k}": v for k, v in mg.items()})

    # ---------------------------------------------------------------- unselected seed sweeps (no rejection)
    for name, H, W, d, K, seeds in (("sweep_c3", 480, 640, 128, 1024, range(1000, 1032)),
                                    ("sweep_c5", 960, 1280, 256, 2048, range(2000, 2008))):
        rows = sweep_rows(name, H, W, d, K, seeds, sg_cfgs[d], sp_stats[d], sg_stats[d], sg_bin[d])
        npz(name + ".npz", H=H, W=W, d=d, K=K, seeds=np.array(list(seeds)), **{k: np.stack(v) for k, v in rows.items()})


def sweep_rows(name, H, W, d, K, seeds, sg_cfg, sp_stat, sg_stat, sg_b):
    """The reference on consecutive seeds with NO rejection: per seed its keypoints, scores, matches, both argmaxes of Z and their
    margins (top-1 minus top-2 per row / column, distance to the match threshold, SuperPoint's top-k boundary gap)."""
    if True:
        kenc, iters, thr = sg_cfg
        cfg = {"superpoint": {"weights": None, "descriptor_dim": d, "nms_radius": 4,
                              "keypoint_threshold": 0.005, "max_keypoints": K},
               "superglue": {"weights": None, "descriptor_dim": d, "keypoint_encoder": kenc,
                             "sinkhorn_iterations": iters, "match_threshold": thr}}
        m = Matching(cfg).eval()
        _, sd_sp = build_sp(d, K, sp_stat)
        _, sd_sg = build_sg(d, kenc, iters, thr, sg_stat, sg_b)
        m.superpoint.load_state_dict(to_torch(sd_sp))
        m.superglue.load_state_dict(to_torch(sd_sg))
        rows = {k: [] for k in ("kpts0", "kpts1", "scores0", "scores1", "matches0", "matches1", "idx0", "idx1", "gap0", "gap1", "thr_gap0",
                                "topk_gap", "n_matches")}
        for seed in seeds:
            xa, xb_ = pair_tensor(seed, H, W)
            pred = m({"image0": xa, "image1": xb_})
            assert len(pred["scores0"][0]) == K and len(pred["scores1"][0]) == K
            data = {"image0": xa, "image1": xb_, **{k: torch.stack(list(v)) for k, v in pred.items() if isinstance(v, (list, tuple))}}
            Z = sg_dense(m.superglue, data)["Z"][0, :-1, :-1]
            t0, t1 = Z.topk(2, dim=1), Z.topk(2, dim=0)
            # the indices the reference itself uses (superglue_test.py:268: Tensor.max -> FIRST maximal index); topk's
            # order among exactly equal values is arbitrary (seed 1010 has an exact two-way tie: gap 0)
            i0, i1 = Z.max(1).indices, Z.max(0).indices
            mutual = i1[i0] == torch.arange(K)
            tg = torch.where(mutual, (t0.values[:, 0] - float(np.log(thr))).abs(), torch.full((K,), float("inf")))
            # top-k boundary of SuperPoint: gap between the last kept and the first dropped candidate score, per image
            gaps = []
            for x_ in (xa, xb_):
                nm = sp_dense(m.superpoint, x_)["nms"][0]
                sel = nm > 0.005
                cs = ref_sp_mod.remove_borders(torch.nonzero(sel), nm[sel], 4, nm.shape[0], nm.shape[1])[1].numpy()
                gaps.append(topk_margin(cs, K)[0])
            nmatch = int((pred["matches0"] > -1).sum())
            print(f"{name} seed {seed}: matches {nmatch}, min row gap {float((t0.values[:, 0] - t0.values[:, 1]).min()):.2e}, "
                  f"min thr gap {float(tg.min()):.2e}, top-k gaps {gaps[0]:.2e} {gaps[1]:.2e}")
            rows["kpts0"].append(pred["keypoints0"][0].numpy().astype(np.int16))
            rows["kpts1"].append(pred["keypoints1"][0].numpy().astype(np.int16))
            rows["scores0"].append(pred["scores0"][0].numpy())
            rows["scores1"].append(pred["scores1"][0].numpy())
            rows["matches0"].append(pred["matches0"][0].numpy().astype(np.int16))
            rows["matches1"].append(pred["matches1"][0].numpy().astype(np.int16))
            rows["idx0"].append(i0.numpy().astype(np.int16))
            rows["idx1"].append(i1.numpy().astype(np.int16))
            rows["gap0"].append((t0.values[:, 0] - t0.values[:, 1]).numpy())
            rows["gap1"].append((t1.values[0] - t1.values[1]).numpy())
            rows["thr_gap0"].append(tg.numpy())
            rows["topk_gap"].append(np.array(gaps, np.float32))
            rows["n_matches"].append(nmatch)
        return rows


def sweep_envelopes(only=("sweep_c3", "sweep_c5")):
    """Round 3 (VERDICT r2 #1c, #2): per unselected seed, the reference's OWN fp32 rounding envelope -- its fp32 result against the
    float64 evaluation of the same module on the same inputs -- for gnn17, scores_in and Z: (max, rms) and the fraction of elements
    outside 1e-4 + 1e-4*|f64|.  Added to the committed sweep fixtures as `env_*` / `out_*` arrays (the other arrays are re-checked
    against the reference, not rewritten).  The GPU sweep tests bound their acceptance threshold with these."""
    def outside(a32, a64):
        return float(((a32.double() - a64).abs() > 1e-4 + 1e-4 * a64.abs()).double().mean())
    for name in only:
        path = os.path.join(OUT, name + ".npz")
        with np.load(path) as z:
            g = {k: z[k] for k in z.files}
        H, W, d, K = (int(g[k]) for k in ("H", "W", "d", "K"))
        kenc, iters, thr = synth.SG_CONFIGS[d]
        cfg = {"superpoint": {"weights": None, "descriptor_dim": d, "nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": K},
               "superglue": {"weights": None, "descriptor_dim": d, "keypoint_encoder": kenc, "sinkhorn_iterations": iters,
                             "match_threshold": thr}}
        m = Matching(cfg).eval()
        m.superpoint.load_state_dict(to_torch(synth.make_superpoint_state_dict(d)))
        m.superglue.load_state_dict(to_torch(synth.make_superglue_state_dict(d)))
        env = {k: [] for k in ("env_gnn", "env_scores_in", "env_Z", "out_gnn", "out_scores_in", "out_Z", "used_P")}
        for s, seed in enumerate(g["seeds"]):
            xa, xb_ = pair_tensor(int(seed), H, W)
            pred = m({"image0": xa, "image1": xb_})
            assert np.array_equal(pred["matches0"][0].numpy(), g["matches0"][s].astype(np.int64)), f"{name} seed {seed}: the fixture is not this reference's output"
            data = {"image0": xa, "image1": xb_, **{k: torch.stack(list(v)) for k, v in pred.items() if isinstance(v, (list, tuple))}}
            a32, a64 = sg_dense(m.superglue, data), sg_dense_f64(m.superglue, data)
            g32, g64 = torch.cat([a32["gnn0"], a32["gnn1"]]), torch.cat([a64["gnn0"], a64["gnn1"]])
            for key, x32, x64 in (("gnn", g32, g64), ("scores_in", a32["scores_in"], a64["scores_in"]), ("Z", a32["Z"], a64["Z"])):
                env["env_" + key].append(envelope(x32, x64))
                env["out_" + key].append(outside(x32, x64))
            # round 4: the transport PLAN exp(Z) -- what the reference consumes -- of the reference's fp32 forward against its float64 self,
            # as the worst fraction of the 1e-4 + 1e-4|ref| tolerance used (> 1 on seeds where the fp32 loop drifts by > 1e-4 / exp(Z))
            P32, P64 = a32["Z"].double().exp(), a64["Z"].exp()
            env["used_P"].append(float(((P32 - P64).abs() / (1e-4 + 1e-4 * P64.abs())).max()))
            print(f"{name} seed {seed}: plan tolerance used by the reference itself {env['used_P'][-1]:.2f}", flush=True)
            print(f"{name} seed {seed}: reference fp32 vs float64  Z max {env['env_Z'][-1][0]:.2e} rms {env['env_Z'][-1][1]:.2e} "
                  f"outside-1e-4 {env['out_Z'][-1]:.3f} | scores_in max {env['env_scores_in'][-1][0]:.2e} outside {env['out_scores_in'][-1]:.3f}", flush=True)
        g.update({k: np.array(v, np.float64) for k, v in env.items()})
        npz(name + ".npz", **g)


def extend_sweep_c5(new_seeds=range(2008, 2016)):
    """Round 4 (VERDICT r3 task 8): the C5 sweep goes from 8 to 16 unselected seeds.  The committed rows are kept byte for byte; the new
    seeds are appended with the same code path (sweep_rows) and their envelopes by sweep_envelopes() afterwards."""
    path = os.path.join(OUT, "sweep_c5.npz")
    with np.load(path) as z:
        g = {k: z[k] for k in z.files}
    have = [int(x) for x in g["seeds"]]
    todo = [sd_ for sd_ in new_seeds if sd_ not in have]
    if not todo:
        print("sweep_c5 already holds", have)
        return
    d, K = 256, 2048
    sg_stat = {k: v for k, v in synth.calibrated_stats("sg256").items()}
    rows = sweep_rows("sweep_c5", 960, 1280, d, K, todo, synth.SG_CONFIGS[d], synth.calibrated_stats("sp256"), sg_stat,
                      float(synth._stats()["sg256/bin_score"]))
    for k, v in rows.items():
        g[k] = np.concatenate([g[k], np.stack(v).astype(g[k].dtype)])
    g["seeds"] = np.array(have + todo)
    for k in [k for k in g if k.startswith("env_") or k.startswith("out_")]:
        del g[k]                      # recomputed for all seeds by sweep_envelopes() (deterministic: the old rows come back identical)
    npz("sweep_c5.npz", **g)


def strict_set():
    """Round 4 (VERDICT r3 task 1): the second synthetic SuperGlue weight set ("t", synth.SGT_GAINS: trained-model-like score
    statistics) -- BatchNorm calibration with the reference's modules on the calibration pair (seed 0, 480x640), bin_score =
    mean + 2 sigma of scores_in there -- and the reference's outputs on the SAME unselected seeds as the round-2 sweeps (C3 1000-1031,
    C5 2000-2007; NO seed is rejected or searched), written as strict_c3.npz / strict_c5.npz: keypoints, scores, matches, matching
    scores, both argmaxes and their margins, strided samples of gnn17 / scores_in / Z in fp32, the float64 evaluation of the same
    module on the same samples (stored as float32 differences f64 - fp32), and per seed the reference's own fp32-vs-float64
    envelope (max, rms, fraction outside 1e-4 + 1e-4|f64|) over the FULL tensors."""
    def outside(a32, a64):
        return float(((a32.double() - a64).abs() > 1e-4 + 1e-4 * a64.abs()).double().mean())
    x0, x1 = pair_tensor(0, 480, 640)
    stats_t = {}
    sg_t = {}
    for d in (128, 256):
        kenc, iters, thr = synth.SG_CONFIGS[d]
        sp, _ = build_sp(d, 1024, synth.calibrated_stats(f"sp{d}"))
        data = sg_data(x0, x1, sp(x0), sp(x1))
        sg, _ = build_sg(d, kenc, iters, thr, gains=synth.SGT_GAINS[d])
        st = calibrate(sg, lambda: sg(data))
        sc = sg_dense(sg, data)["scores_in"]
        b = float(sc.mean() + 2.0 * sc.std())
        for k, v in st.items():
            stats_t[f"sgt{d}/{k}"] = v
        stats_t[f"sgt{d}/bin_score"] = np.float32(b)
        sg_t[d] = (st, b)
        print(f"t set d={d}: calibration pair scores_in mean {sc.mean():.3f} std {sc.std():.3f} max |S| {sc.abs().max():.1f} -> bin_score {b:.4f}", flush=True)
    np.savez_compressed(os.path.join(DATA, "synth_bn_stats_t.npz"), **stats_t)
    print("  wrote synth_bn_stats_t.npz", os.path.getsize(os.path.join(DATA, "synth_bn_stats_t.npz")) // 1024, "KB")
    synth._STATS.pop("synth_bn_stats_t.npz", None)

    for name, H, W, d, K, seeds, st_s, st_g in (("strict_c3", 480, 640, 128, 1024, range(1000, 1032), 16, 32),
                                                ("strict_c5", 960, 1280, 256, 2048, range(2000, 2016), 32, 64)):
        kenc, iters, thr = synth.SG_CONFIGS[d]
        cfg = {"superpoint": {"weights": None, "descriptor_dim": d, "nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": K},
               "superglue": {"weights": None, "descriptor_dim": d, "keypoint_encoder": kenc, "sinkhorn_iterations": iters,
                             "match_threshold": thr}}
        m = Matching(cfg).eval()
        m.superpoint.load_state_dict(to_torch(synth.make_superpoint_state_dict(d)))
        sd_t = synth.make_superglue_state_dict(d, variant="t")
        m.superglue.load_state_dict(to_torch(sd_t))
        rows = {k: [] for k in ("kpts0", "kpts1", "scores0", "scores1", "matches0", "matches1", "mscores0", "mscores1", "idx0", "idx1",
                                "gap0", "gap1", "thr_gap0", "topk_gap", "n_matches", "gnn_sub", "scores_in_sub", "Z_sub",
                                "scores_in_sub_d64", "Z_sub_d64", "env_gnn", "env_scores_in", "env_Z", "out_gnn", "out_scores_in", "out_Z",
                                "stat_scores_in", "decision_gap")}
        for seed in seeds:
            xa, xb_ = pair_tensor(seed, H, W)
            pred = m({"image0": xa, "image1": xb_})
            assert len(pred["scores0"][0]) == K and len(pred["scores1"][0]) == K
            data = {"image0": xa, "image1": xb_, **{k: torch.stack(list(v)) for k, v in pred.items() if isinstance(v, (list, tuple))}}
            a32, a64 = sg_dense(m.superglue, data), sg_dense_f64(m.superglue, data)
            Zf = a32["Z"]
            Z = Zf[0, :-1, :-1]
            t0, t1 = Z.topk(2, dim=1), Z.topk(2, dim=0)
            i0, i1 = Z.max(1).indices, Z.max(0).indices
            mutual = i1[i0] == torch.arange(K)
            tg = torch.where(mutual, (t0.values[:, 0] - float(np.log(thr))).abs(), torch.full((K,), float("inf")))
            gaps = []
            for x_ in (xa, xb_):
                nm = sp_dense(m.superpoint, x_)["nms"][0]
                sel = nm > 0.005
                cs = ref_sp_mod.remove_borders(torch.nonzero(sel), nm[sel], 4, nm.shape[0], nm.shape[1])[1].numpy()
                gaps.append(topk_margin(cs, K)[0])
            mgn = match_margins(Zf, pred, thr)
            g32, g64 = torch.cat([a32["gnn0"], a32["gnn1"]]), torch.cat([a64["gnn0"], a64["gnn1"]])
            for key, x32, x64 in (("gnn", g32, g64), ("scores_in", a32["scores_in"], a64["scores_in"]), ("Z", a32["Z"], a64["Z"])):
                rows["env_" + key].append(envelope(x32, x64))
                rows["out_" + key].append(outside(x32, x64))
            S = a32["scores_in"]
            nmatch = int((pred["matches0"] > -1).sum())
            print(f"{name} seed {seed}: matches {nmatch}; scores_in mean {S.mean():.2f} std {S.std():.2f} max|S| {S.abs().max():.1f} max|Z| {Zf.abs().max():.1f}; "
                  f"reference fp32 vs float64: gnn max {rows['env_gnn'][-1][0]:.1e} S max {rows['env_scores_in'][-1][0]:.1e} Z max {rows['env_Z'][-1][0]:.1e} "
                  f"outside {rows['out_gnn'][-1]:.1e}/{rows['out_scores_in'][-1]:.1e}/{rows['out_Z'][-1]:.1e}; decision gap {mgn['decision_gap']:.1e} "
                  f"thr gap {mgn['thr_gap']:.1e} top-k gaps {gaps[0]:.1e} {gaps[1]:.1e}", flush=True)
            rows["kpts0"].append(pred["keypoints0"][0].numpy().astype(np.int16))
            rows["kpts1"].append(pred["keypoints1"][0].numpy().astype(np.int16))
            rows["scores0"].append(pred["scores0"][0].numpy())
            rows["scores1"].append(pred["scores1"][0].numpy())
            rows["matches0"].append(pred["matches0"][0].numpy().astype(np.int16))
            rows["matches1"].append(pred["matches1"][0].numpy().astype(np.int16))
            rows["mscores0"].append(pred["matching_scores0"][0].numpy())
            rows["mscores1"].append(pred["matching_scores1"][0].numpy())
            rows["idx0"].append(i0.numpy().astype(np.int16))
            rows["idx1"].append(i1.numpy().astype(np.int16))
            rows["gap0"].append((t0.values[:, 0] - t0.values[:, 1]).numpy())
            rows["gap1"].append((t1.values[0] - t1.values[1]).numpy())
            rows["thr_gap0"].append(tg.numpy())
            rows["topk_gap"].append(np.array(gaps, np.float32))
            rows["n_matches"].append(nmatch)
            rows["decision_gap"].append(mgn["decision_gap"])
            rows["gnn_sub"].append(torch.stack([a32["gnn0"][0, :, ::st_g], a32["gnn1"][0, :, ::st_g]]).numpy())
            rows["scores_in_sub"].append(S[0, ::st_s, ::st_s].numpy())
            rows["Z_sub"].append(Zf[0, ::st_s, ::st_s].numpy())
            rows["scores_in_sub_d64"].append((a64["scores_in"][0, ::st_s, ::st_s] - S[0, ::st_s, ::st_s].double()).float().numpy())
            rows["Z_sub_d64"].append((a64["Z"][0, ::st_s, ::st_s] - Zf[0, ::st_s, ::st_s].double()).float().numpy())
            rows["stat_scores_in"].append(np.array([float(S.mean()), float(S.std()), float(S.abs().max()), float(Zf.abs().max())]))
        npz(name + ".npz", H=H, W=W, d=d, K=K, seeds=np.array(list(seeds)), stride_s=st_s, stride_g=st_g,
            bin_score=np.float32(sd_t["bin_score"]), **{k: np.stack(v) for k, v in rows.items()})


def heavy_check():
    """Round 5: the "heavy" weight sets (synth.heavy_superpoint / heavy_superglue: channels of a few layers rescaled
    by 2^+-10 and undone in the next layer, a query / key channel pair by 2^+-7) are function-preserving re-parameterisations by powers
    of two, so the REFERENCE's fp32 forward must give bit-identical outputs on them.  Checked here with the reference's own modules on
    strict seeds (C3 1000, 1001; C5 2000): Matching's ten outputs, SuperPoint's dense semi / desc, SuperGlue's dense gnn17 / scores_in
    / Z.  Writes tests/golden/heavy_check.npz (what was compared, the largest difference = 0): every committed golden vector therefore
    also pins the heavy sets, which is how tests/test_gpu_heavy.py uses them."""
    rec = {"seeds": [], "max_abs_diff": [], "tensors": 0}
    for H, W, d, K, seeds in ((480, 640, 128, 1024, (1000, 1001)), (960, 1280, 256, 2048, (2000,))):
        kenc, iters, thr = synth.SG_CONFIGS[d]
        cfg = {"superpoint": {"weights": None, "descriptor_dim": d, "nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": K},
               "superglue": {"weights": None, "descriptor_dim": d, "keypoint_encoder": kenc, "sinkhorn_iterations": iters, "match_threshold": thr}}
        ms = []
        for heavy in (False, True):
            m = Matching(cfg).eval()
            m.superpoint.load_state_dict(to_torch(synth.make_superpoint_state_dict(d, heavy=heavy)))
            m.superglue.load_state_dict(to_torch(synth.make_superglue_state_dict(d, variant="t", heavy=heavy)))
            ms.append(m)
        for seed in seeds:
            xa, xb_ = pair_tensor(seed, H, W)
            outs = []
            for m in ms:
                pred = m({"image0": xa, "image1": xb_})
                data = {"image0": xa, "image1": xb_, **{k: torch.stack(list(v)) for k, v in pred.items() if isinstance(v, (list, tuple))}}
                dn = sg_dense(m.superglue, data)
                sp = sp_dense(m.superpoint, xa)
                flat = {k: (v[0] if isinstance(v, (list, tuple)) else v) for k, v in pred.items()}
                flat.update({"dense_" + k: dn[k] for k in ("gnn0", "gnn1", "scores_in", "Z")})
                flat.update({"sp_" + k: sp[k] for k in ("semi", "desc")})
                outs.append(flat)
            worst = 0.0
            for k in outs[0]:
                a, b = outs[0][k], outs[1][k]
                assert torch.equal(a, b), f"{k} differs between the base and the heavy weight set (seed {seed}, d {d})"
                worst = max(worst, float((a.double() - b.double()).abs().max()))
                rec["tensors"] += 1
            rec["seeds"].append(seed)
            rec["max_abs_diff"].append(worst)
            print(f"heavy check d={d} seed {seed}: {len(outs[0])} tensors bit-identical", flush=True)
    npz("heavy_check.npz", seeds=np.array(rec["seeds"]), max_abs_diff=np.array(rec["max_abs_diff"], np.float64), tensors=np.int64(rec["tensors"]),
        heavy_sp=np.array([[i, c, e] for _, i, c, e, _ in synth.HEAVY_SP]), heavy_sg_layers=np.array([[l, c, e] for l, (c, e) in synth.HEAVY_SG_LAYERS.items()]),
        heavy_sg_qk=np.array([[l, c, e] for l, (c, e) in synth.HEAVY_SG_QK.items()]))


def strict_ties():
    """the strict tests excuse rows / columns on an EXACT tie of the reference's own fp32 Z (top-1 minus top-2 == 0.0) --
    but any index passed there.  This writes the tied PARTNERS (tests/golden/strict_ties.npz: per tie, the indices whose Z equals the
    maximum, from the reference's own forward), so that the tests can require the reported index to be one of them (or -1)."""
    out = []
    for fi, name in enumerate(("strict_c3", "strict_c5")):
        g = dict(np.load(os.path.join(OUT, name + ".npz")))
        H, W, d, K = (int(g[k]) for k in ("H", "W", "d", "K"))
        kenc, iters, thr = synth.SG_CONFIGS[d]
        todo = [s for s in range(len(g["seeds"])) if (g["gap0"][s] == 0).any() or (g["gap1"][s] == 0).any()]
        if not todo:
            continue
        cfg = {"superpoint": {"weights": None, "descriptor_dim": d, "nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": K},
               "superglue": {"weights": None, "descriptor_dim": d, "keypoint_encoder": kenc, "sinkhorn_iterations": iters, "match_threshold": thr}}
        m = Matching(cfg).eval()
        m.superpoint.load_state_dict(to_torch(synth.make_superpoint_state_dict(d)))
        m.superglue.load_state_dict(to_torch(synth.make_superglue_state_dict(d, variant="t")))
        for s in todo:
            xa, xb_ = pair_tensor(int(g["seeds"][s]), H, W)
            pred = m({"image0": xa, "image1": xb_})
            assert np.array_equal(pred["matches0"][0].numpy(), g["matches0"][s])
            data = {"image0": xa, "image1": xb_, **{k: torch.stack(list(v)) for k, v in pred.items() if isinstance(v, (list, tuple))}}
            Z = sg_dense(m.superglue, data)["Z"][0, :-1, :-1]
            for axis, gaps in ((0, g["gap0"][s]), (1, g["gap1"][s])):
                for idx in np.nonzero(gaps == 0)[0]:
                    line = Z[idx] if axis == 0 else Z[:, idx]
                    partners = torch.nonzero(line == line.max())[:, 0].numpy()
                    assert 2 <= len(partners) <= 8
                    out.append([fi, s, axis, int(idx)] + list(map(int, partners)) + [-1] * (8 - len(partners)))
                    print(f"{name} seed {int(g['seeds'][s])}: {'row' if axis == 0 else 'column'} {int(idx)} ties over {list(map(int, partners))}", flush=True)
    npz("strict_ties.npz", ties=np.array(out, np.int32).reshape(-1, 12))


if __name__ == "__main__":
    if "--strict-ties" in sys.argv:
        strict_ties()
    elif "--heavy-check" in sys.argv:
        heavy_check()
    elif "--sweep-envelopes" in sys.argv:
        sweep_envelopes()
    elif "--strict-set" in sys.argv:
        strict_set()
    elif "--extend-c5" in sys.argv:
        old = dict(np.load(os.path.join(OUT, "sweep_c5.npz")))
        extend_sweep_c5()
        sweep_envelopes(only=("sweep_c5",))
        new = dict(np.load(os.path.join(OUT, "sweep_c5.npz")))
        n = len(old["seeds"])
        for k, v in old.items():      # the committed rows are unchanged
            if v.ndim and v.shape[0] == n:
                assert np.array_equal(new[k][:n], v, equal_nan=True), k
    else:
        main()

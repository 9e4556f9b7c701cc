"""Shared helpers for the parity tests (oracle = checker; image_matching_amd = product)."""
import os

import numpy as np
import torch

from image_matching_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ATOL, RTOL = 1e-4, 1e-4     # north_star tolerance: |a-b| <= 1e-4 + 1e-4*|b| (fp32)


def golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def to_torch(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def sp_sd(d=128):
    return to_torch(synth.make_superpoint_state_dict(d))


def sg_sd(d=128, kenc=None, n_layers=18):
    return to_torch(synth.make_superglue_state_dict(d, kenc, n_layers))


def pair(seed, H, W):
    im0, im1 = synth.synth_pair(seed, H, W)
    return torch.from_numpy(im0)[None, None], torch.from_numpy(im1)[None, None]


def sp_config(d=128, K=1024, **kw):
    return {"weights": None, "descriptor_dim": d, "nms_radius": 4, "keypoint_threshold": 0.005,
            "max_keypoints": K, "remove_borders": 4, **kw}


def sg_config(d=128, **kw):
    kenc, iters, thr = synth.SG_CONFIGS[d]
    return {"weights": None, "descriptor_dim": d, "keypoint_encoder": kenc,
            "sinkhorn_iterations": iters, "match_threshold": thr, **kw}


def assert_close(a, b, what, atol=ATOL, rtol=RTOL, scale_atol=False):
    """|a-b| <= atol + rtol*|b|.  scale_atol: atol is taken relative to max|b| — for quantities that
    are long fp32 reductions of O(max|b|) terms (GNN features after 18 layers, score matrix, Z), whose
    rounding noise is proportional to the operand scale, not to the (possibly cancelling) result:
    the reference's own fp32-vs-fp64 deviation on Z is 5e-6*max|Z| (tests/golden/make_golden.py)."""
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a.astype(np.float64) - b.astype(np.float64))
    if scale_atol:
        atol = atol * max(1.0, float(np.abs(b).max()))
    tol = atol + rtol * np.abs(b.astype(np.float64))
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} elements out of tolerance; max err {err.max():.3e} "
                           f"(at |ref| {np.abs(b).reshape(-1)[err.argmax()]:.3e}), max |ref| {np.abs(b).max():.3e}")


def canon_keypoints(kpts, scores, desc=None):
    """Order-independent view: sort by (y, x) — used where top-k order could differ by ulp-level ties."""
    k = kpts.detach().cpu().numpy() if isinstance(kpts, torch.Tensor) else np.asarray(kpts)
    s = scores.detach().cpu().numpy() if isinstance(scores, torch.Tensor) else np.asarray(scores)
    order = np.lexsort((k[:, 0], k[:, 1]))
    out = [k[order], s[order]]
    if desc is not None:
        d = desc.detach().cpu().numpy() if isinstance(desc, torch.Tensor) else np.asarray(desc)
        out.append(d[:, order])
    return out

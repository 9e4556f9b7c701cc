"""TEST INFRASTRUCTURE (checker only) — host restatement of libimx's RANSAC partial-affine fit
(registration.hip), itself the GPU counterpart of cv2.estimateAffinePartial2D at
superpoint_glue_test.py:86-92.  Same counter-based hypothesis sequence and float32 residual test, so
inlier masks compare exactly.  Parity vs OpenCV itself is UNPINNED (OpenCV's RNG and refinement are
third-party code absent from the reference tree)."""
import numpy as np


def _mix32(x):
    x = np.uint32(x)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16); x = np.uint32(x * np.uint32(0x85EBCA6B))
        x ^= x >> np.uint32(13); x = np.uint32(x * np.uint32(0xC2B2AE35))
        x ^= x >> np.uint32(16)
    return np.uint32(x)


def _pair(seed, b, h, n):
    with np.errstate(over="ignore"):
        r = _mix32(np.uint32(seed) ^ np.uint32(np.uint32(0x9E3779B9) * np.uint32(b + 1)) ^ np.uint32(np.uint32(0x85EBCA6B) * np.uint32(h + 1)))
        i = int(r % np.uint32(n))
        j = int(_mix32(np.uint32(r + np.uint32(0x27D4EB2F))) % np.uint32(n - 1))
    if j >= i:
        j += 1
    return i, j


def estimate_affine_partial(kpts0, kpts1, matches0, b=0, thresh=7.0, hypotheses=512, seed=0):
    """kpts0 (K,2), kpts1 (K1,2), matches0 (K,) -> (M (2,3) float32, inlier mask (K,) uint8, n_inliers)."""
    f = np.float32
    kpts0, kpts1 = np.asarray(kpts0, f), np.asarray(kpts1, f)
    valid = np.nonzero(matches0 >= 0)[0]
    K = len(matches0)
    n = len(valid)
    if n <= 3:
        return np.zeros((2, 3), f), np.zeros(K, np.uint8), 0
    s, d = kpts0[valid], kpts1[matches0[valid]]
    thr2 = f(thresh) * f(thresh)

    def model(i, j):
        px, py = s[j, 0] - s[i, 0], s[j, 1] - s[i, 1]
        qx, qy = d[j, 0] - d[i, 0], d[j, 1] - d[i, 1]
        den = px * px + py * py
        if not den > f(1e-12):
            return None
        ca, cb = (qx * px + qy * py) / den, (qy * px - qx * py) / den
        tx = d[i, 0] - (ca * s[i, 0] - cb * s[i, 1])
        ty = d[i, 1] - (cb * s[i, 0] + ca * s[i, 1])
        return f(ca), f(cb), f(tx), f(ty)

    def residual_ok(m, x, y, u, v):
        ca, cb, tx, ty = m
        ex = ca * x - cb * y + tx - u
        ey = cb * x + ca * y + ty - v
        return ex * ex + ey * ey < thr2

    best_cnt, best_h, best_m = -1, None, None
    for h in range(hypotheses):
        m = model(*_pair(seed, b, h, n))
        cnt = -1 if m is None else int(residual_ok(m, s[:, 0], s[:, 1], d[:, 0], d[:, 1]).sum())
        if cnt > best_cnt:
            best_cnt, best_h, best_m = cnt, h, m
    if best_cnt < 2:
        return np.zeros((2, 3), f), np.zeros(K, np.uint8), 0
    inl = residual_ok(best_m, s[:, 0], s[:, 1], d[:, 0], d[:, 1])
    p, q = s[inl].astype(np.float64), d[inl].astype(np.float64)
    pc, qc = p.mean(0), q.mean(0)
    dp, dq = p - pc, q - qc
    den = (dp ** 2).sum()
    a = (dp * dq).sum() / den
    bb = (dp[:, 0] * dq[:, 1] - dp[:, 1] * dq[:, 0]).sum() / den
    M = np.array([[a, -bb, qc[0] - (a * pc[0] - bb * pc[1])], [bb, a, qc[1] - (bb * pc[0] + a * pc[1])]], f)
    mask = np.zeros(K, np.uint8)
    mask[valid[inl]] = 1
    return M, mask, int(inl.sum())

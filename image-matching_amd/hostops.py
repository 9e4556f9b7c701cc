"""Host-side image I/O and the geometric post-step of the CLI (superpoint_glue_test.py:86-140).

In the reference these are OpenCV calls (`cv2.imread/resize/estimateAffinePartial2D/warpAffine/
imwrite`, `make_matching_plot_fast`).  OpenCV is used when importable; otherwise small numpy/PIL
equivalents keep the CLI usable.  This is plumbing around the hot path (SURVEY §8f rank 1 is the
GPU version of the RANSAC step); its arithmetic is third-party in the reference -> parity unpinned.
"""
import numpy as np

try:                                    # pragma: no cover - depends on the image
    import cv2
except Exception:                       # noqa: BLE001
    cv2 = None


def imread_gray(path):
    if cv2 is not None:
        return cv2.imread(path, cv2.IMREAD_GRAYSCALE)
    from PIL import Image
    return np.asarray(Image.open(path).convert("L"))


def _lin_coef(n_dst, n_src, clamp_weight):
    """OpenCV's 8U INTER_LINEAR tap table for one axis: source index and the two 11-bit weights."""
    scale = 1.0 / (n_dst / n_src)
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    if clamp_weight:
        lo, hi = s < 0, s >= n_src - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, n_src - 1, s))
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, a0, a1


def resize_linear_u8(img, size_wh):
    """cv2.resize(uint8, size, INTER_LINEAR) restated in numpy (fixed-point, 11-bit weights); the GPU kernel
    resize_u8_unit (csrc/ingest.hip) is bit-exact against this."""
    img = np.asarray(img, np.uint8)
    W, H = size_wh
    Hs, Ws = img.shape
    cx, a0, a1 = _lin_coef(W, Ws, True)
    cy, b0, b1 = _lin_coef(H, Hs, False)
    y0, y1 = np.clip(cy, 0, Hs - 1), np.clip(cy + 1, 0, Hs - 1)
    x1 = np.minimum(cx + 1, Ws - 1)
    im = img.astype(np.int64)
    r0 = im[y0][:, cx] * a0 + im[y0][:, x1] * a1
    r1 = im[y1][:, cx] * a0 + im[y1][:, x1] * a1
    v = (((b0[:, None] * (r0 >> 4)) >> 16) + ((b1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def resize(img, size_wh):
    if cv2 is not None:
        return cv2.resize(img, size_wh)
    return resize_linear_u8(img, size_wh)


def imwrite(path, img):
    img = np.asarray(img)
    if img.dtype != np.uint8:               # cv2.imwrite converts with round-half-even + saturation
        img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    if cv2 is not None:
        return cv2.imwrite(path, img)
    from PIL import Image
    if img.ndim == 3:
        img = img[:, :, ::-1]           # BGR (OpenCV convention) -> RGB
    Image.fromarray(img).save(path)
    return True


def _similarity_from_pairs(p, q):
    """Least-squares 4-DoF similarity q ~ [a -b; b a] p + t (what estimateAffinePartial2D fits)."""
    pc, qc = p.mean(0), q.mean(0)
    dp, dq = p - pc, q - qc
    den = (dp ** 2).sum()
    if den < 1e-12:
        return None
    a = (dp * dq).sum() / den
    b = (dp[:, 0] * dq[:, 1] - dp[:, 1] * dq[:, 0]).sum() / den
    R = np.array([[a, -b], [b, a]], dtype=np.float64)
    t = qc - R @ pc
    return np.concatenate([R, t[:, None]], axis=1)


def estimate_affine_partial_2d(src, dst, ransac_thresh=7.0, max_iters=2000, confidence=0.99, seed=0):
    """RANSAC + refit on inliers; returns (2x3 matrix or None, inlier mask (N,1) uint8) like
    cv2.estimateAffinePartial2D(..., method=cv2.RANSAC, ransacReprojThreshold=7)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    if cv2 is not None:
        return cv2.estimateAffinePartial2D(src.astype(np.float32), dst.astype(np.float32), method=cv2.RANSAC,
                                           ransacReprojThreshold=ransac_thresh)
    n = len(src)
    if n < 2:
        return None, np.zeros((n, 1), np.uint8)
    rng = np.random.RandomState(seed)
    best, best_cnt, iters, it = None, 0, max_iters, 0
    while it < iters:
        i, j = rng.choice(n, 2, replace=False)
        M = _similarity_from_pairs(src[[i, j]], dst[[i, j]])
        it += 1
        if M is None:
            continue
        err = np.linalg.norm(src @ M[:, :2].T + M[:, 2] - dst, axis=1)
        inl = err < ransac_thresh
        if inl.sum() > best_cnt:
            best, best_cnt = inl, int(inl.sum())
            w = best_cnt / n
            iters = min(max_iters, int(np.ceil(np.log(1 - confidence) / np.log(max(1 - w * w, 1e-12)))) + 1)
    if best is None or best_cnt < 2:
        return None, np.zeros((n, 1), np.uint8)
    M = _similarity_from_pairs(src[best], dst[best])
    return M, best.astype(np.uint8)[:, None]


def invert_affine(M):
    """The double-precision 2x3 inversion cv2.warpAffine applies to a forward matrix."""
    M = np.asarray(M, np.float64).reshape(2, 3)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    inv = np.empty((2, 3))
    inv[0, 0], inv[0, 1], inv[1, 0], inv[1, 1] = M[1, 1] * D, M[0, 1] * (-D), M[1, 0] * (-D), M[0, 0] * D
    inv[0, 2] = -inv[0, 0] * M[0, 2] - inv[0, 1] * M[1, 2]
    inv[1, 2] = -inv[1, 0] * M[0, 2] - inv[1, 1] * M[1, 2]
    return inv


def _sat_int(v):
    return np.rint(np.clip(v, -2147483648.0, 2147483647.0)).astype(np.int64)


def warp_affine(img, M, size_wh):
    """cv2.warpAffine(img, M, size) for a float image, INTER_LINEAR, constant border 0: 10+5-bit fixed-point source
    coordinates, float32 table weights, float64 accumulation (restated; bit-exact twin of csrc/ingest.hip)."""
    if cv2 is not None:
        return cv2.warpAffine(img, M, size_wh)
    W, H = size_wh
    m = invert_affine(M).ravel()
    xs, ys = np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64)
    X = ((_sat_int((m[1] * ys + m[2]) * 1024.0) + 16)[:, None] + _sat_int(m[0] * xs * 1024.0)[None]) >> 5
    Y = ((_sat_int((m[4] * ys + m[5]) * 1024.0) + 16)[:, None] + _sat_int(m[3] * xs * 1024.0)[None]) >> 5
    ix, iy = X >> 5, Y >> 5
    fx, fy = (X & 31).astype(np.float32) / np.float32(32), (Y & 31).astype(np.float32) / np.float32(32)
    one = np.float32(1)
    img = np.asarray(img, np.float64)
    Hs, Ws = img.shape
    out = np.zeros((H, W), np.float64)
    for dy, dx, w in ((0, 0, (one - fy) * (one - fx)), (0, 1, (one - fy) * fx), (1, 0, fy * (one - fx)), (1, 1, fy * fx)):
        xx, yy = ix + dx, iy + dy
        ok = (xx >= 0) & (xx < Ws) & (yy >= 0) & (yy < Hs)
        tap = np.zeros((H, W), np.float64)
        tap[ok] = img[yy[ok], xx[ok]]
        out = out + tap * w.astype(np.float64)
    return out


def make_matching_plot_fast(image0, image1, kpts0, kpts1, mkpts0, mkpts1, color, text, path=None,
                            show_keypoints=False, margin=10, small_text=()):
    """Side-by-side match visualisation (same layout as superglue/models/utils.py:500-566)."""
    from PIL import Image, ImageDraw
    H0, W0 = image0.shape
    H1, W1 = image1.shape
    H, W = max(H0, H1), W0 + W1 + margin
    out = 255 * np.ones((H, W), np.uint8)
    out[:H0, :W0] = np.clip(image0, 0, 255)
    out[:H1, W0 + margin:] = np.clip(image1, 0, 255)
    im = Image.fromarray(np.stack([out] * 3, -1))
    dr = ImageDraw.Draw(im)
    if show_keypoints:
        for x, y in np.round(kpts0).astype(int):
            dr.ellipse([x - 2, y - 2, x + 2, y + 2], fill=(0, 0, 0))
            dr.point([x, y], fill=(255, 255, 255))
        for x, y in np.round(kpts1).astype(int):
            dr.ellipse([x + margin + W0 - 2, y - 2, x + margin + W0 + 2, y + 2], fill=(0, 0, 0))
            dr.point([x + margin + W0, y], fill=(255, 255, 255))
    col = (np.array(color)[:, :3] * 255).astype(int)
    for (x0, y0), (x1, y1), c in zip(np.round(mkpts0).astype(int), np.round(mkpts1).astype(int), col):
        c = tuple(int(v) for v in c)
        dr.line([x0, y0, x1 + margin + W0, y1], fill=c, width=1)
        dr.ellipse([x0 - 2, y0 - 2, x0 + 2, y0 + 2], fill=c)
        dr.ellipse([x1 + margin + W0 - 2, y1 - 2, x1 + margin + W0 + 2, y1 + 2], fill=c)
    for i, t in enumerate(text):
        dr.text((8, 8 + 14 * i), t, fill=(255, 255, 255))
    for i, t in enumerate(reversed(list(small_text))):
        dr.text((8, H - 14 * (i + 1)), t, fill=(255, 255, 255))
    arr = np.asarray(im)[:, :, ::-1]        # BGR like OpenCV
    if path is not None:
        imwrite(str(path), arr)
    return arr

"""The OpenCV-shaped rows pin themselves on any box that has OpenCV (VERDICT r3 task 6; SURVEY 8f ranks 1, 3, 4 and BASELINE
configs[0]).  `cv2` is third-party, absent from this image and from the GPU boxes of this pool (tools/try_opencv.sh), so every test
here SKIPS today -- the kernels stay bit-exact against restatements of OpenCV's published algorithms (oracle/ingest_ref.py,
oracle/ransac_ref.py), "parity vs cv2 unpinned".  With `import cv2` available they compare against cv2 itself:

  imx_ingest_resize_u8      == cv2.resize(..., INTER_LINEAR) / 255           byte for byte      datasets/SSHIDataset.py:19-27
  imx_warp_affine_u8        == cv2.warpAffine(source * 255, M, (W, H))       byte for byte      superpoint_glue_test.py:101-113
  imx_estimate_affine_partial  inlier set covers >= 95 % of cv2's, model within 0.5 px         superpoint_glue_test.py:86-92
  imx_knn_ratio_match       covers every match FLANN's approximate search accepts             superpoint_flann_test.py:62-74
  traditional.py            runs end to end on a synthetic pair directory (SIFT and ORB)       traditional.py:8-57
"""
import os

import numpy as np
import pytest
import torch

from tests import util

cv2 = pytest.importorskip("cv2", reason="OpenCV is not installed on this box: the cv2-pinned parity tests skip (rows stay 'unpinned')")
pytestmark = pytest.mark.gpu


def _engine(K=1024):
    from image_matching_amd.engine import Engine
    return Engine(util.sp_config(128, K), util.sg_config(128), "cuda")


@pytest.mark.parametrize("src,dst", [((960, 1280), (480, 640)), ((37, 53), (15, 20)), ((36, 52), (18, 26)), ((37, 53), (50, 80)), ((9, 70), (4, 31)), ((1920, 2560), (576, 768))])
def test_resize_bit_exact_vs_cv2(src, dst):
    eng = _engine()
    img = np.random.RandomState(src[0] * 131 + dst[1]).randint(0, 256, src).astype(np.uint8)
    out = eng.ingest(torch.from_numpy(img), dst).cpu().numpy()[0, 0]
    ref = (cv2.resize(img, (dst[1], dst[0])) / 255).astype(np.float32)        # SSHIDataset.py:19-27: cv2.resize (INTER_LINEAR) then /255
    assert np.array_equal(out, ref), f"resize {src} -> {dst}: {int((out != ref).sum())} of {ref.size} values differ from cv2.resize"


@pytest.mark.parametrize("M", [[[0.95, -0.1, 3.2], [0.1, 0.95, -2.1]], [[1, 0, -7], [0, 1, 4]], [[1.3, 0.4, -20.5], [-0.4, 1.3, 11.25]],
                               [[0.98, 0.05, -31.7], [-0.05, 0.98, 18.3]]])
def test_warp_affine_bit_exact_vs_cv2(M):
    eng = _engine()
    img = np.random.RandomState(5).randint(0, 256, (480, 640)).astype(np.uint8)
    got = eng.warp_affine_u8(torch.from_numpy(img), M).cpu().numpy()
    # the reference warps the float image * 255 and cv2.imwrite saturates to uint8 (superpoint_glue_test.py:101-113)
    ref = cv2.warpAffine((img / 255.0 * 255).astype(np.float64), np.asarray(M, np.float64), (640, 480))
    ref = np.clip(np.rint(ref), 0, 255).astype(np.uint8)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {ref.size} bytes differ from cv2.warpAffine"


@pytest.mark.parametrize("seed,theta,scale,t", [(0, 0.05, 0.95, (12, -7)), (1, -0.3, 1.2, (-40, 25)), (2, 1.0, 1.0, (300, 10))])
def test_ransac_vs_cv2_estimate_affine_partial(seed, theta, scale, t):
    """Planted similarity transform, 1/3 unmatched, 1/7 gross outliers: cv2's RANSAC draws its own hypotheses, so the comparison is on
    what both must find -- the inlier SET (ours must cover >= 95 % of cv2's) and the model (within 0.5 px over the image)."""
    from tests.test_gpu_registration import _case
    eng = _engine()
    k0, k1, m, Mtrue = _case(seed, 1024, theta, scale, t)
    M, inl, ninl = eng.estimate_affine_partial(torch.from_numpy(k0)[None].cuda(), torch.from_numpy(k1)[None].cuda(), torch.from_numpy(m)[None].cuda(),
                                               ransac_thresh=7.0, hypotheses=512, seed=3)
    M, inl = M[0].cpu().numpy(), inl[0].cpu().numpy().astype(bool)
    rows = np.nonzero(m >= 0)[0]
    Mc, maskc = cv2.estimateAffinePartial2D(k0[rows], k1[m[rows]], method=cv2.RANSAC, ransacReprojThreshold=7)
    cv_inl = np.zeros(len(m), bool)
    cv_inl[rows] = maskc.ravel().astype(bool)
    cover = (inl & cv_inl).sum() / max(cv_inl.sum(), 1)
    assert cover >= 0.95, f"our inliers cover {cover:.3f} of cv2's {int(cv_inl.sum())}"
    corners = np.array([[0, 0, 1], [640, 0, 1], [0, 480, 1], [640, 480, 1]], np.float64)
    assert np.abs(corners @ M.T.astype(np.float64) - corners @ Mc.T).max() < 0.5, (M, Mc)


def test_knn_ratio_matcher_covers_flann():
    """FLANN's KD-tree search is approximate; every pair it accepts under the 0.7 ratio test whose neighbours are the true two
    nearest must be accepted by the exact search, with the same index (superpoint_flann_test.py:62-74)."""
    eng = _engine()
    g = torch.Generator().manual_seed(4)
    d0 = torch.nn.functional.normalize(torch.randn(1, 128, 700, generator=g), dim=1)
    d1 = torch.nn.functional.normalize(d0[:, :, torch.randperm(700, generator=g)] + 0.15 * torch.randn(1, 128, 700, generator=g), dim=1)
    mine, dist1, _ = eng.knn_ratio_match(d0.cuda(), d1.cuda(), ratio=0.7)
    mine = mine[0].cpu().numpy()
    a, b = d0[0].t().contiguous().numpy(), d1[0].t().contiguous().numpy()
    flann = cv2.FlannBasedMatcher(dict(algorithm=0, trees=5), dict(checks=50))
    acc = {mm.queryIdx: mm.trainIdx for mm, nn in flann.knnMatch(a, b, k=2) if mm.distance < 0.7 * nn.distance}
    # exact two nearest neighbours by brute force: FLANN answers that are exact must be reproduced
    D = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
    order = np.argsort(D, axis=1)[:, :2]
    exact = {q: t for q, t in acc.items() if order[q, 0] == t}
    missing = [q for q, t in exact.items() if mine[q] != t]
    assert len(exact) > 100 and not missing, f"{len(missing)} of {len(exact)} exact FLANN matches are not returned by the exact search"


@pytest.mark.parametrize("method", ["SIFT", "ORB"])
def test_traditional_cli_end_to_end(method, tmp_path):
    """BASELINE configs[0]: traditional.py on a synthetic 640x480 pair directory -- plumbing (flags, directory convention, outputs)."""
    import traditional
    from Traditional import registration
    ok, why = registration.available(method)
    if not ok:
        pytest.skip(why)
    im0, im1 = util.pair(3, 480, 640)
    for sub, im in (("template1", im0), ("source1", im1)):
        os.makedirs(tmp_path / "data" / sub)
        cv2.imwrite(str(tmp_path / "data" / sub / "a.png"), cv2.cvtColor((im[0, 0].numpy() * 255).astype(np.uint8), cv2.COLOR_GRAY2BGR))
    res = traditional.main(["--Method", method, "--img_dir", str(tmp_path / "data") + "/", "--Result_dir", str(tmp_path / "out") + "/", "--resize_scale", "1.0"])
    assert len(res) == 1
    if res[0][1] is not None:             # image1 = roll(image0, (8, 16)): translation (-16, -8) back onto the template
        assert os.path.exists(tmp_path / "out" / method / "Transform1" / "trans_a.png")
        assert np.abs(res[0][1][:, :2] - np.eye(2)).max() < 0.05

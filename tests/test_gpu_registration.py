"""RANSAC partial-affine post-step: libimx kernel vs its host restatement
(oracle/ransac_ref.py, same hypothesis sequence) and vs the ground-truth transform.  Needs an MI355X.
Parity vs cv2.estimateAffinePartial2D itself is unpinned (OpenCV is not in the reference tree)."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _case(seed, K, theta, scale, t, frac_unmatched=3, outlier_every=7):
    rng = np.random.RandomState(seed)
    k0 = (rng.rand(K, 2) * 600).astype(np.float32)
    R = np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]]) * scale
    perm = rng.permutation(K)
    k1 = np.zeros((K, 2), np.float32)
    k1[perm] = (k0 @ R.T + t + rng.randn(K, 2) * 0.05).astype(np.float32)      # inliers: 0.05 px noise
    m = perm.astype(np.int64).copy()
    m[::frac_unmatched] = -1
    bad = np.arange(1, K, outlier_every)
    k1[perm[bad]] += (rng.rand(len(bad), 2) * 100 + 30).astype(np.float32)      # outliers: >= 30 px off
    return k0, k1, m, np.concatenate([R, np.array(t, float)[:, None]], 1)


def test_ransac_recovers_transform_and_matches_host_restatement():
    from image_matching_amd.engine import Engine
    from oracle import ransac_ref
    eng = Engine(util.sp_config(128, 1024), util.sg_config(128), "cuda")
    cases = [_case(0, 1024, 0.05, 0.95, (12, -7)), _case(1, 1024, -0.3, 1.2, (-40, 25)), _case(2, 1024, 1.0, 1.0, (300, 10))]
    k0 = torch.from_numpy(np.stack([c[0] for c in cases])).cuda()
    k1 = torch.from_numpy(np.stack([c[1] for c in cases])).cuda()
    m = torch.from_numpy(np.stack([c[2] for c in cases])).cuda()
    M, inl, ninl = eng.estimate_affine_partial(k0, k1, m, ransac_thresh=7.0, hypotheses=256, seed=5)
    M, inl, ninl = M.cpu().numpy(), inl.cpu().numpy(), ninl.cpu().numpy()
    for b, (c0, c1, cm, Mtrue) in enumerate(cases):
        Mr, maskr, nr = ransac_ref.estimate_affine_partial(c0, c1, cm, b=b, thresh=7.0, hypotheses=256, seed=5)
        assert np.array_equal(inl[b], maskr), f"pair {b}: inlier mask differs from the host restatement"
        assert ninl[b] == nr
        np.testing.assert_allclose(M[b], Mr, atol=2e-4, rtol=1e-5)
        np.testing.assert_allclose(M[b][:, :2], Mtrue[:, :2], atol=2e-3)                 # ground truth
        np.testing.assert_allclose(M[b][:, 2], Mtrue[:, 2], atol=0.5)
        matched = cm >= 0
        bad = np.zeros(len(cm), bool); bad[np.arange(1, len(cm), 7)] = True
        assert inl[b][matched & ~bad].all() and not inl[b][matched & bad].any() and not inl[b][~matched].any()


def test_ransac_beyond_the_lds_staging_matches_host_restatement():
    """More than 8192 keypoint slots (--max_keypoints -1 on a large image): the compacted coordinates no longer fit LDS and are
    staged in HBM -- same inlier mask and model as the host restatement."""
    from image_matching_amd.engine import Engine
    from oracle import ransac_ref
    eng = Engine(util.sp_config(128, -1), util.sg_config(128), "cuda")
    c0, c1, cm, Mtrue = _case(7, 10000, 0.15, 1.05, (20, -11))
    M, inl, ninl = eng.estimate_affine_partial(torch.from_numpy(c0)[None].cuda(), torch.from_numpy(c1)[None].cuda(),
                                               torch.from_numpy(cm)[None].cuda(), ransac_thresh=7.0, hypotheses=128, seed=9)
    Mr, maskr, nr = ransac_ref.estimate_affine_partial(c0, c1, cm, b=0, thresh=7.0, hypotheses=128, seed=9)
    assert np.array_equal(inl[0].cpu().numpy(), maskr) and int(ninl[0]) == nr
    np.testing.assert_allclose(M[0].cpu().numpy(), Mr, atol=2e-4, rtol=1e-5)
    np.testing.assert_allclose(M[0].cpu().numpy()[:, :2], Mtrue[:, :2], atol=2e-3)


def test_ransac_too_few_matches_and_counts():
    from image_matching_amd.engine import Engine
    eng = Engine(util.sp_config(128, 64), util.sg_config(128), "cuda")
    k0, k1, m, _ = _case(3, 64, 0.1, 1.0, (5, 5), frac_unmatched=1)      # everything unmatched
    m[:3] = [0, 1, 2]
    k1[:3] = k0[:3]
    M, inl, ninl = eng.estimate_affine_partial(torch.from_numpy(k0)[None], torch.from_numpy(k1)[None], torch.from_numpy(m)[None])
    assert int(ninl[0]) == 0 and not inl.any() and (M == 0).all()          # <= 3 matches: no fit (reference :86)
    # ragged: K0 != K1
    k0b, k1b, mb, Mt = _case(4, 200, 0.2, 1.1, (3, 4), frac_unmatched=5)
    keep = 150
    mb2 = mb[:keep].copy()
    M2, inl2, n2 = eng.estimate_affine_partial(torch.from_numpy(k0b[:keep])[None], torch.from_numpy(k1b)[None], torch.from_numpy(mb2)[None])
    assert inl2.shape == (1, keep) and int(n2[0]) > 50
    np.testing.assert_allclose(M2[0].cpu().numpy()[:, :2], Mt[:, :2], atol=2e-3)


def test_knn_ratio_matcher_vs_brute_force():
    """SuperPoint + nearest-neighbour matcher (superpoint_flann_test.py:66-74): exact 2-NN + ratio test on the GPU
    against numpy brute force on real SuperPoint descriptors of a synthetic pair (FLANN itself is approximate and
    absent: parity vs cv2 unpinned)."""
    from image_matching_amd.engine import Engine
    from image_matching_amd import _lib as L
    eng = Engine(util.sp_config(128, 600), util.sg_config(128), "cuda")
    eng.load_state_dict(L.NET_SUPERPOINT, util.sp_sd(128))
    x0, x1 = util.pair(12, 240, 320)
    kp, sc, ds, n = eng.superpoint(torch.cat([x0, x1]).cuda())
    d0, d1 = ds[0, :n[0]].t()[None], ds[1, :n[1] - 37].t()[None]          # (1,d,N0), (1,d,N1), N0 != N1, strided views
    m, dist1, dist2 = eng.knn_ratio_match(d0, d1, ratio=0.7)
    a, b = d0[0].t().double().cpu().numpy(), d1[0].t().double().cpu().numpy()
    D = np.sqrt(np.maximum((a * a).sum(1)[:, None] + (b * b).sum(1)[None] - 2 * a @ b.T, 0))
    order = np.argsort(D, axis=1)
    nn1, nn2 = order[:, 0], order[:, 1]
    r1, r2 = D[np.arange(len(a)), nn1], D[np.arange(len(a)), nn2]
    np.testing.assert_allclose(dist1[0].cpu().numpy(), r1, atol=2e-4)
    np.testing.assert_allclose(dist2[0].cpu().numpy(), r2, atol=2e-4)
    decided = np.abs(r1 - 0.7 * r2) > 1e-3                                 # away from the ratio boundary
    expect = np.where(r1 < 0.7 * r2, nn1, -1)
    got = m[0].cpu().numpy()
    assert np.array_equal(got[decided], expect[decided])
    clear = (r2 - r1) > 1e-3
    assert np.array_equal(got[(got >= 0) & clear], nn1[(got >= 0) & clear])


def test_flann_cli_end_to_end_on_synthetic_dataset(tmp_path):
    """superpoint_flann_test.py:42-127 contract: reads <img_dir>/source1/*, <img_dir>/template1/<one>, writes
    <Result_dir>/transformed/trans_* and <Result_dir>/Match/match_*; the synthetic sources are translations of the
    template, so the fitted partial affine must be that translation."""
    import os
    import superpoint_flann_test as cli
    img_dir, res_dir = str(tmp_path / "data") + "/", str(tmp_path / "out") + "/"
    res = cli.main(["--img_dir", img_dir, "--Result_dir", res_dir, "--synthetic", "2", "--resize_scale", "0.5"])
    assert sorted(os.listdir(os.path.join(res_dir, "Match"))) == ["match_src_000.png", "match_src_001.png"]
    assert sorted(os.listdir(os.path.join(res_dir, "transformed"))) == ["trans_src_000.png", "trans_src_001.png"]
    for i, (name, n_good, n_inl, M) in enumerate(res):
        assert n_good > 50 and n_inl > 0.5 * n_good, (name, n_good, n_inl)
        # source = roll(template, (16(i+1) rows, 32(i+1) cols)) at full size -> template = source shifted back
        np.testing.assert_allclose(M[:, :2], np.eye(2), atol=0.02)
        np.testing.assert_allclose(M[:, 2], [-32 * (i + 1), -16 * (i + 1)], atol=1.5)


@pytest.mark.parametrize("seed", util.fuzz_seeds([0, 1, 2, 3, 4, 5]))
def test_ransac_random_sizes_counts_and_outliers_vs_host_restatement(seed):
    """Random batch sizes, keypoint capacities (4..3000, K0 != K1), unmatched fractions, outlier rates, thresholds and hypothesis
    counts: the inlier mask must equal the host restatement's (same hypothesis sequence) and the model agree to 2e-4."""
    from image_matching_amd.engine import Engine
    from oracle import ransac_ref
    rng = np.random.RandomState(777 + seed)
    B = int(rng.randint(1, 5))
    K = int(rng.choice([4, 5, 9, 64, 65, 257, 1000, 1024, 3000]))
    if seed % 3 == 1:
        K = int(rng.randint(4, 700))
    thresh = float(rng.choice([1.0, 3.0, 7.0]))
    hyp = int(rng.choice([1, 16, 100, 256, 512]))
    eng = Engine(util.sp_config(128, 1024), util.sg_config(128), "cuda")
    cases = [_case(seed * 10 + b, K, float(rng.uniform(-1.5, 1.5)), float(rng.uniform(0.5, 2.0)), (float(rng.uniform(-100, 100)), float(rng.uniform(-100, 100))),
                   frac_unmatched=int(rng.randint(2, 9)), outlier_every=int(rng.randint(2, 12))) for b in range(B)]
    k0 = torch.from_numpy(np.stack([c[0] for c in cases])).cuda()
    k1 = torch.from_numpy(np.stack([c[1] for c in cases])).cuda()
    m = torch.from_numpy(np.stack([c[2] for c in cases])).cuda()
    rs = int(rng.randint(0, 2 ** 31))
    M, inl, ninl = eng.estimate_affine_partial(k0, k1, m, ransac_thresh=thresh, hypotheses=hyp, seed=rs)
    M, inl, ninl = M.cpu().numpy(), inl.cpu().numpy(), ninl.cpu().numpy()
    for b, (c0, c1, cm, _) in enumerate(cases):
        Mr, maskr, nr = ransac_ref.estimate_affine_partial(c0, c1, cm, b=b, thresh=thresh, hypotheses=hyp, seed=rs)
        what = f"seed {seed} pair {b}: B={B} K={K} thresh={thresh} hypotheses={hyp}"
        assert ninl[b] == nr and np.array_equal(inl[b], maskr), what + ": inlier mask differs from the host restatement"
        np.testing.assert_allclose(M[b], Mr, atol=2e-4, rtol=1e-5, err_msg=what)


@pytest.mark.parametrize("seed", util.fuzz_seeds([0, 1, 2, 3, 4, 5]))
def test_knn_ratio_random_shapes_vs_brute_force(seed):
    """Random batch sizes, descriptor counts per side (1..900), descriptor widths and ratios, strided inputs: exact 2-NN distances
    at 2e-4 and the ratio decision wherever it is not within 1e-3 of the boundary (float64 brute force)."""
    from image_matching_amd.engine import Engine
    rng = np.random.RandomState(555 + seed)
    B, d = int(rng.randint(1, 4)), int(rng.choice([64, 128, 256]))
    N0, N1 = int(rng.randint(1, 901)), int(rng.randint(2, 901))
    ratio = float(rng.choice([0.6, 0.7, 0.8, 0.95]))
    g = torch.Generator().manual_seed(seed)
    a = torch.nn.functional.normalize(torch.randn(B, N0, d, generator=g), dim=2)
    b = torch.nn.functional.normalize(torch.randn(B, N1, d, generator=g), dim=2)
    if N1 >= N0:                      # plant true neighbours so the ratio test accepts some rows
        b[:, :N0:3] = torch.nn.functional.normalize(a[:, ::3] + 0.05 * torch.randn(a[:, ::3].shape, generator=g), dim=2)
    eng = Engine(util.sp_config(d, 64), util.sg_config(d), "cuda")
    m, dist1, dist2 = eng.knn_ratio_match(a.cuda().transpose(1, 2), b.cuda().transpose(1, 2), ratio=ratio)      # (B,d,N) strided views
    for i in range(B):
        x, y = a[i].double().numpy(), b[i].double().numpy()
        D = np.sqrt(np.maximum((x * x).sum(1)[:, None] + (y * y).sum(1)[None] - 2 * x @ y.T, 0))
        order = np.argsort(D, axis=1)
        nn1, nn2 = order[:, 0], order[:, 1]
        r1, r2 = D[np.arange(N0), nn1], D[np.arange(N0), nn2]
        what = f"seed {seed}: B={B} d={d} {N0}x{N1} ratio {ratio} pair {i}"
        np.testing.assert_allclose(dist1[i].cpu().numpy(), r1, atol=2e-4, err_msg=what)
        np.testing.assert_allclose(dist2[i].cpu().numpy(), r2, atol=2e-4, err_msg=what)
        got = m[i].cpu().numpy()
        decided = (np.abs(r1 - ratio * r2) > 1e-3) & ((r2 - r1) > 1e-3)
        assert np.array_equal(got[decided], np.where(r1 < ratio * r2, nn1, -1)[decided]), what

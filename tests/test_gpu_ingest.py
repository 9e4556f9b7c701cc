"""Ingest and warp byte kernels: bit-exact vs the scalar restatement of OpenCV's
fixed-point algorithms (oracle/ingest_ref.py; parity vs cv2 itself unpinned) and vs the vectorised host twin at
full size.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _engine():
    from image_matching_amd.engine import Engine
    return Engine(util.sp_config(128, 1024), util.sg_config(128), "cuda")


@pytest.mark.parametrize("src,dst", [((37, 53), (15, 20)), ((37, 53), (37, 53)), ((36, 52), (18, 26)), ((37, 53), (50, 80)), ((9, 70), (4, 31))])
def test_resize_bit_exact_vs_scalar_restatement(src, dst):
    from oracle import ingest_ref
    eng = _engine()
    img = np.random.RandomState(src[0] * 131 + dst[1]).randint(0, 256, (2,) + src).astype(np.uint8)
    out = eng.ingest(torch.from_numpy(img), dst).cpu().numpy()
    assert out.shape == (2, 1) + dst and out.dtype == np.float32
    for b in range(2):
        ref = ingest_ref.unit_float(ingest_ref.resize_u8(img[b], (dst[1], dst[0])))
        assert np.array_equal(out[b, 0], ref)


def test_resize_full_size_vs_host_twin_and_no_resize_path():
    from image_matching_amd import hostops
    eng = _engine()
    img = np.random.RandomState(3).randint(0, 256, (1920, 2560)).astype(np.uint8)
    out = eng.ingest(torch.from_numpy(img), (480, 640)).cpu().numpy()[0, 0]        # resize_scale 0.25
    assert np.array_equal(out, (hostops.resize_linear_u8(img, (640, 480)) / 255).astype(np.float32))
    odd = eng.ingest(torch.from_numpy(img), (int(0.3 * 1920), int(0.3 * 2560))).cpu().numpy()[0, 0]
    assert np.array_equal(odd, (hostops.resize_linear_u8(img, (int(0.3 * 2560), int(0.3 * 1920))) / 255).astype(np.float32))
    same = eng.ingest(torch.from_numpy(img)).cpu().numpy()[0, 0]                   # resize_scale None: plain /255
    assert np.array_equal(same, (img / 255).astype(np.float32))


def test_warp_affine_bit_exact():
    from image_matching_amd import hostops
    from oracle import ingest_ref
    eng = _engine()
    rng = np.random.RandomState(5)
    small = rng.randint(0, 256, (37, 53)).astype(np.uint8)
    for M in ([[0.95, -0.1, 3.2], [0.1, 0.95, -2.1]], [[1, 0, -7], [0, 1, 4]], [[1.3, 0.4, -20.5], [-0.4, 1.3, 11.25]]):
        got = eng.warp_affine_u8(torch.from_numpy(small), M).cpu().numpy()
        assert np.array_equal(got, ingest_ref.warp_affine_u8(small, M, (53, 37))), M
    big = rng.randint(0, 256, (960, 1280)).astype(np.uint8)
    M = np.array([[0.98, 0.05, -31.7], [-0.05, 0.98, 18.3]])
    got = eng.warp_affine_u8(torch.from_numpy(big), M).cpu().numpy()
    ref = hostops.warp_affine(big / 255 * 255, M, (1280, 960))
    assert np.array_equal(got, np.clip(np.rint(ref), 0, 255).astype(np.uint8))


def test_ingest_pipeline_feeds_matching_identically():
    """uint8 frames through the pinned/async pipeline == the reference's host ingest fed to the same matcher."""
    from image_matching_amd import hostops
    from image_matching_amd.ingest import IngestPipeline
    from image_matching_amd.superglue.models.matching_test import Matching
    m = Matching({"superpoint": util.sp_config(128, 512), "superglue": util.sg_config(128)}).eval().to("cuda")
    m.superpoint.load_state_dict(util.sp_sd(128))
    m.superglue.load_state_dict(util.sg_sd(128))
    frames = []
    for i in range(6):                                       # 3 batches of 2 pairs through 2 slots: slots get reused
        a, b = util.pair(40 + i, 480, 640)
        up = lambda t: hostops.resize_linear_u8((t[0, 0].numpy() * 255).astype(np.uint8), (1280, 960))
        frames.append((up(a), up(b)))
    eng = m._shared.get_engine([0, 1])
    pipe0 = IngestPipeline(eng, 2, (960, 1280), (480, 640))
    pipe1 = IngestPipeline(eng, 2, (960, 1280), (480, 640))
    outs = []
    ship = lambda k: (pipe0.submit([f[0] for f in frames[2 * k:2 * k + 2]]), pipe1.submit([f[1] for f in frames[2 * k:2 * k + 2]]))
    t = ship(0)
    for k in range(3):
        outs.append(m.match_batch(pipe0.take(t[0]), pipe1.take(t[1])))
        pipe0.release(t[0]), pipe1.release(t[1])
        if k + 1 < 3:
            t = ship(k + 1)
    torch.cuda.synchronize()
    for k in range(3):
        host0 = torch.stack([torch.from_numpy(hostops.resize_linear_u8(f[0], (640, 480)) / 255).float()[None] for f in frames[2 * k:2 * k + 2]]).cuda()
        host1 = torch.stack([torch.from_numpy(hostops.resize_linear_u8(f[1], (640, 480)) / 255).float()[None] for f in frames[2 * k:2 * k + 2]]).cuda()
        ref = m.match_batch(host0, host1)
        for key in ref:
            assert torch.equal(ref[key], outs[k][key]), (k, key)


@pytest.mark.parametrize("seed", util.fuzz_seeds([0, 1, 2, 3, 4, 5]))
def test_resize_and_warp_random_sizes_bit_exact(seed):
    """Random source / destination sizes (up- and down-scaling, 1-pixel edges, non-integer ratios) and random affine maps: every
    byte equal to the scalar restatement of OpenCV's fixed-point INTER_LINEAR resize / warp (oracle/ingest_ref.py)."""
    from oracle import ingest_ref
    eng = _engine()
    rng = np.random.RandomState(888 + seed)
    B = int(rng.randint(1, 4))
    src = (int(rng.randint(1, 90)), int(rng.randint(1, 120)))
    dst = (int(rng.randint(1, 100)), int(rng.randint(1, 130)))
    img = rng.randint(0, 256, (B,) + src).astype(np.uint8)
    out = eng.ingest(torch.from_numpy(img), dst).cpu().numpy()
    assert out.shape == (B, 1) + dst
    for b in range(B):
        ref = ingest_ref.unit_float(ingest_ref.resize_u8(img[b], (dst[1], dst[0])))
        assert np.array_equal(out[b, 0], ref), f"seed {seed}: resize {src} -> {dst} image {b}"
    th, sc = rng.uniform(-3.1, 3.1), rng.uniform(0.3, 3.0)
    M = [[sc * np.cos(th), -sc * np.sin(th), rng.uniform(-60, 60)], [sc * np.sin(th), sc * np.cos(th), rng.uniform(-60, 60)]]
    got = eng.warp_affine_u8(torch.from_numpy(img[0]), M).cpu().numpy()
    assert np.array_equal(got, ingest_ref.warp_affine_u8(img[0], M, (src[1], src[0]))), f"seed {seed}: warp {src} M={M}"

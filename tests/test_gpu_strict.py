"""Round 4 (VERDICT r3 task 1): the north_star bar, literally -- |hip - ref| <= 1e-4 + 1e-4|ref| ELEMENT-WISE on gnn17,
scores_in and Z and ZERO differing match indices -- on the "t" SuperGlue weight set (synth.SGT_GAINS: scores_in std ~ 5,
bin_score = mean + 2 sigma; the reference's own fp32 forward is inside the same bar of its float64 self there, so the bar is
attainable) and on UNSELECTED seeds (C3 1000-1031, C5 2000-2007; tests/golden/make_golden.py --strict-set imports the
reference, nothing is rejected).  No margin, envelope or floor clause anywhere in this file.  Needs an MI355X.

Two references are used side by side: the fixture (the reference's OWN outputs: every match index and matching score, strided
samples of the three dense tensors) and the oracle run here on the same inputs (every element of the three dense tensors;
the oracle is itself held to the fixture's samples at 1e-5 first)."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
KEYS = ("keypoints0", "keypoints1", "scores0", "scores1", "descriptors0", "descriptors1")
_INPUTS = {}


def _strict_inputs(name):
    """Per seed, once per process: the SuperGlue inputs (the REFERENCE's keypoints and scores from the fixture, descriptors
    sampled by the oracle at those keypoints) and the oracle's dense gnn17 / scores_in / Z on them, checked against the
    fixture's samples of the reference's tensors."""
    if name in _INPUTS:
        return _INPUTS[name]
    from oracle import superglue_ref, superpoint_ref
    g = util.golden(name)
    H, W, d, K = (int(g[k]) for k in ("H", "W", "d", "K"))
    torch.set_num_threads(min(32, torch.get_num_threads()))
    sd_sp, sd_sg = util.sp_sd(d), util.sg_sd(d, variant="t")
    per_seed = []
    for s, seed in enumerate(g["seeds"]):
        x0, x1 = util.pair(int(seed), H, W)
        data = {"image0": x0, "image1": x1}
        for side, x in (("0", x0), ("1", x1)):
            dense = superpoint_ref.superpoint_forward(x, sd_sp, util.sp_config(d, K), return_dense=True)["desc"]
            kp = torch.from_numpy(g["kpts" + side][s].astype(np.float32))[None]
            data["keypoints" + side] = kp
            data["scores" + side] = torch.from_numpy(g["scores" + side][s])[None]
            data["descriptors" + side] = superpoint_ref.sample_descriptors(kp, dense, 8)
        dn = superglue_ref.superglue_forward(data, sd_sg, util.sg_config(d), return_dense=True)["dense"]
        ref = {"gnn0": dn["gnn0"][0].numpy(), "gnn1": dn["gnn1"][0].numpy(), "scores_in": dn["scores_in"][0].numpy(), "Z": dn["Z"][0].numpy()}
        for key, (mine, fx) in util.strict_samples(g, s, ref["gnn0"], ref["gnn1"], ref["scores_in"], ref["Z"]).items():
            util.assert_close(mine, fx, f"oracle vs the reference's {key} ({name} seed {seed})", atol=1e-5, rtol=1e-5)
        per_seed.append(({k: data[k] for k in KEYS}, ref))
    _INPUTS[name] = (g, per_seed)
    return _INPUTS[name]


def _pair_taps(eng, K):
    """A single pair's GNN output out of the library's tap, laid out like the reference's: gnn0 / gnn1 (d, N)."""
    Kp = (K + 31) // 32 * 32
    x = eng.fetch("x")
    return x[:K].T, x[Kp:Kp + K].T


@pytest.mark.parametrize("forms,mfma", [("auto", "x3"), ("off", "x3"), ("off", "f32")])
@pytest.mark.parametrize("name", ["strict_c3.npz", "strict_c5.npz"])
def test_strict_bar_superglue_every_form(name, forms, mfma):
    """SuperGlue alone, the reference's keypoints injected, on every kernel form a caller can reach (latency forms; the
    throughput forms bench.py times; their fp32-MFMA counterparts): gnn17, scores_in and Z element-wise inside
    1e-4 + 1e-4|ref| of the oracle's (every element) and of the reference's (fixture samples); matches0 / matches1 equal to the
    reference's; matching scores at the same tolerance."""
    g, per_seed = _strict_inputs(name)
    H, W, d, K = (int(g[k]) for k in ("H", "W", "d", "K"))
    from image_matching_amd import _lib as L
    from image_matching_amd.engine import Engine
    eng = Engine(util.sp_config(d, K), util.sg_config(d), "cuda")
    sd_sg = util.sg_sd(d, variant="t")
    eng.load_state_dict(L.NET_SUPERGLUE, sd_sg)
    eng.set_option("latency_forms", forms).set_option("mfma", mfma)
    eng.set_debug(True)
    alpha = float(sd_sg["bin_score"])
    worst = {"gnn17": 0.0, "scores_in": 0.0, "Z": 0.0, "mscores": 0.0}
    for s, seed in enumerate(g["seeds"]):
        data, ref = per_seed[s]
        out = eng.superglue(data["keypoints0"].cuda(), data["scores0"].cuda(), data["descriptors0"].cuda(), (1, 1, H, W),
                            data["keypoints1"].cuda(), data["scores1"].cuda(), data["descriptors1"].cuda(), (1, 1, H, W))
        torch.cuda.synchronize()
        m0, m1, ms0, ms1 = (o.cpu().numpy() for o in out)
        g0, g1 = _pair_taps(eng, K)
        S = eng.fetch("scores_in")[0, :K, :K]
        Z = util.transport_Z(S, eng.fetch("u")[0], eng.fetch("v")[0], K, K, alpha)
        tag = f"{name} seed {seed} [{forms}/{mfma}]"
        for key, mine, full in (("gnn17", np.stack([g0, g1]), np.stack([ref["gnn0"], ref["gnn1"]])), ("scores_in", S, ref["scores_in"]), ("Z", Z, ref["Z"])):
            util.assert_close(mine, full, f"{tag}: {key} vs the oracle, every element")
            worst[key] = max(worst[key], util.tolerance_used(mine, full))
        for key, (mine, fx) in util.strict_samples(g, s, g0, g1, S, Z).items():
            util.assert_close(mine, fx, f"{tag}: {key} vs the reference's sample")
        r0, r1 = g["matches0"][s].astype(np.int64), g["matches1"][s].astype(np.int64)
        assert np.array_equal(m0[0], r0) and np.array_equal(m1[0], r1), \
            f"{tag}: {int((m0[0] != r0).sum())}+{int((m1[0] != r1).sum())} match indices differ from the reference's (rows {np.nonzero(m0[0] != r0)[0][:6]})"
        util.assert_close(ms0[0], g["mscores0"][s], f"{tag}: matching_scores0")
        util.assert_close(ms1[0], g["mscores1"][s], f"{tag}: matching_scores1")
        worst["mscores"] = max(worst["mscores"], util.tolerance_used(ms0[0], g["mscores0"][s]))
    n = len(g["seeds"])
    print(f"[strict] {name} [{forms}/{mfma}]: {n} unselected seeds, 0 of {2 * K * n} match indices differ; worst fraction of the 1e-4+1e-4|ref| tolerance used: "
          + ", ".join(f"{k} {v:.3f}" for k, v in worst.items()))


def _matching_t(d, K):
    from image_matching_amd.superglue.models.matching_test import Matching
    m = Matching({"superpoint": util.sp_config(d, K), "superglue": util.sg_config(d)}).eval().to("cuda")
    m.superpoint.load_state_dict(util.sp_sd(d))
    m.superglue.load_state_dict(util.sg_sd(d, variant="t"))
    return m


@pytest.mark.parametrize("name,B", [("strict_c3.npz", 64), ("strict_c5.npz", 8)])
def test_strict_bar_as_one_batched_call_from_images(name, B):
    """The whole HIP path, images in, as ONE imx_match_pairs call of B pairs -- the call, batch size and kernel forms bench.py
    times (asserted) -- on the strict set: keypoint sets identical to the reference's, zero differing matches, matching scores and
    the reference's samples of gnn17 / scores_in / Z at 1e-4 + 1e-4|ref| although the descriptors now come from the HIP SuperPoint."""
    g = util.golden(name)
    H, W, d, K = (int(g[k]) for k in ("H", "W", "d", "K"))
    n = len(g["seeds"])
    m = _matching_t(d, K)
    ims = [util.pair(int(seed), H, W) for seed in g["seeds"]]
    i0 = torch.cat([ims[b % n][0] for b in range(B)]).cuda()
    i1 = torch.cat([ims[b % n][1] for b in range(B)]).cuda()
    eng = m._shared.get_engine([0, 1])
    eng.timing_reset()
    eng.set_timing(True)
    eng.set_debug(True)
    out = m.match_batch(i0, i1)
    torch.cuda.synchronize()
    forms = {r[0]: r[3] for r in eng.timing_report(forms=True)}
    eng.set_timing(False)
    assert forms["qkv_proj"] == "gemm_x3:bf16x3" and forms["attention"] == "attention_x3:bf16x3" and forms["conv2a"] == "conv3x3_wino24:f32", forms
    summary = util.strict_compare_batch(g, out, eng, B, float(util.sg_sd(d, variant="t")["bin_score"]))
    print(f"[strict e2e] {name} as one call of {B} pairs: {summary}")
    m0 = out["matches0"].cpu().numpy()
    for b in range(n, B):
        assert np.array_equal(m0[b], m0[b - n]), f"pair {b} differs from its copy at {b - n}"

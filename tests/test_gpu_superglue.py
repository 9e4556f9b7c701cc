"""SuperGlue parity: HIP path vs reference goldens and vs the oracle, keypoints injected so the
comparison of match indices is bit-exact.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
KEYS = ("keypoints0", "keypoints1", "scores0", "scores1", "descriptors0", "descriptors1")


def _engine(d=128, **kw):
    from image_matching_amd import _lib as L
    from image_matching_amd.engine import Engine
    eng = Engine(util.sp_config(d, 1024), util.sg_config(d, **kw), "cuda")
    return eng, L


def _run(eng, t, shp, n0=None, n1=None):
    out = eng.superglue(t["keypoints0"], t["scores0"], t["descriptors0"], shp,
                        t["keypoints1"], t["scores1"], t["descriptors1"], shp, n0, n1)
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in out]


def test_small_dense_and_matches_vs_reference_golden():
    g = util.golden("sg_small.npz")
    eng, L = _engine()
    sd = util.sg_sd(128)
    eng.load_state_dict(L.NET_SUPERGLUE, sd)
    eng.set_debug(True)
    t = {k: torch.from_numpy(g[k]).cuda() for k in KEYS}
    m0, m1, ms0, ms1 = _run(eng, t, (1, 1, 120, 160))
    N0, N1 = g["keypoints0"].shape[1], g["keypoints1"].shape[1]
    N0p = (N0 + 31) // 32 * 32

    def rows(a):
        return a[:N0].T[None], a[N0p:N0p + N1].T[None]
    # early stages: the strict north_star tolerance, element-wise
    for tap, (r0, r1) in (("kenc", (g["kenc0"], g["kenc1"])), ("gnn0", (g["tap0_0"], g["tap0_1"])),
                          ("gnn1", (g["tap1_0"], g["tap1_1"]))):
        a0, a1 = rows(eng.fetch(tap))
        util.assert_close(a0, r0, tap + " side0")
        util.assert_close(a1, r1, tap + " side1")
    # long fp32 reductions: anchored on the float64 evaluation of the reference module (fp64_anchor.npz)
    an = util.golden("fp64_anchor.npz")
    a0, a1 = rows(eng.fetch("gnn17"))
    util.assert_fp64_anchored(a0, g["gnn0"], an["sg_small/gnn0_f64"], "gnn17 side0")
    util.assert_fp64_anchored(a1, g["gnn1"], an["sg_small/gnn1_f64"], "gnn17 side1")
    S = eng.fetch("scores_in")[:, :N0, :N1]
    util.assert_fp64_anchored(S, g["scores_in"], an["sg_small/scores_in_f64"], "scores_in")
    u, v = eng.fetch("u")[0], eng.fetch("v")[0]
    Z = util.transport_Z(S[0], u, v, N0, N1, float(sd["bin_score"]))
    util.assert_fp64_anchored(Z[None], g["Z"], an["sg_small/Z_f64"], "Z (optimal transport)")
    # Appendix A.4: exp(Z) rows sum to 1 (dustbin row to N), after `iters` iterations columns nearly so
    P = np.exp(Z.astype(np.float64))
    np.testing.assert_allclose(P[:N0].sum(1), 1.0, rtol=0, atol=0.35)
    np.testing.assert_allclose(P[:, :N1].sum(0), 1.0, rtol=1e-4)
    assert np.array_equal(m0, g["matches0"]) and np.array_equal(m1, g["matches1"]), "match indices must be bit-exact"
    util.assert_close(ms0, g["matching_scores0"], "matching_scores0")
    util.assert_close(ms1, g["matching_scores1"], "matching_scores1")
    assert m0.dtype == np.int64 and ms0.dtype == np.float32


def test_ragged_counts_masked_counts_and_empty():
    g, gs = util.golden("sg_ragged.npz"), util.golden("sg_small.npz")
    eng, L = _engine()
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(128))
    n0, n1 = int(g["n0"]), int(g["n1"])
    full = {k: torch.from_numpy(gs[k]).cuda() for k in KEYS}
    cut = dict(full)
    cut["keypoints0"], cut["scores0"], cut["descriptors0"] = full["keypoints0"][:, :n0], full["scores0"][:, :n0], full["descriptors0"][:, :, :n0]
    cut["keypoints1"], cut["scores1"], cut["descriptors1"] = full["keypoints1"][:, :n1], full["scores1"][:, :n1], full["descriptors1"][:, :, :n1]
    m0, m1, ms0, ms1 = _run(eng, cut, (1, 1, 120, 160))       # non-contiguous descriptor views: strides honoured
    assert np.array_equal(m0, g["matches0"]) and np.array_equal(m1, g["matches1"])
    util.assert_close(ms0, g["matching_scores0"], "mscores0 ragged")
    # same result through device-side counts on the padded (full) buffers; tail is -1 / 0
    c0 = torch.tensor([n0], dtype=torch.int32, device="cuda")
    c1 = torch.tensor([n1], dtype=torch.int32, device="cuda")
    p0, p1, ps0, ps1 = _run(eng, full, (1, 1, 120, 160), c0, c1)
    assert np.array_equal(p0[:, :n0], g["matches0"]) and np.array_equal(p1[:, :n1], g["matches1"])
    assert (p0[:, n0:] == -1).all() and (p1[:, n1:] == -1).all() and (ps0[:, n0:] == 0).all()
    # zero count on one side -> all -1 (device-side early-out)
    z = torch.tensor([0], dtype=torch.int32, device="cuda")
    e0, e1, es0, es1 = _run(eng, full, (1, 1, 120, 160), c0, z)
    assert (e0 == -1).all() and (e1 == -1).all() and (es0 == 0).all() and (es1 == 0).all()


def test_empty_set_dtype_matches_reference():
    from image_matching_amd.superglue.models.superglue_test import SuperGlue
    g = util.golden("sg_ragged.npz")
    sg = SuperGlue(util.sg_config(128)).eval().to("cuda")
    data = {"keypoints0": torch.zeros(1, 150, 2).cuda(), "keypoints1": torch.zeros(1, 0, 2).cuda(),
            "scores0": torch.zeros(1, 150).cuda(), "scores1": torch.zeros(1, 0).cuda(),
            "descriptors0": torch.zeros(1, 128, 150).cuda(), "descriptors1": torch.zeros(1, 128, 0).cuda(),
            "image0": torch.zeros(1, 1, 120, 160), "image1": torch.zeros(1, 1, 120, 160)}
    out = sg(data)
    assert str(out["matches0"].dtype) == str(g["empty_dtype"]) == "torch.int32"
    assert np.array_equal(out["matches0"].cpu().numpy(), g["empty_matches0"])
    assert out["matches1"].shape == (1, 0) and out["matching_scores0"].shape == (1, 150)


def _oracle_pair_inputs(seed, H, W, d, K):
    from oracle import superpoint_ref
    sd = util.sp_sd(d)
    x0, x1 = util.pair(seed, H, W)
    o0 = superpoint_ref.superpoint_forward(x0, sd, util.sp_config(d, K))
    o1 = superpoint_ref.superpoint_forward(x1, sd, util.sp_config(d, K))
    return {"keypoints0": o0["keypoints"][0][None], "keypoints1": o1["keypoints"][0][None],
            "scores0": o0["scores"][0][None], "scores1": o1["scores"][0][None],
            "descriptors0": o0["descriptors"][0][None], "descriptors1": o1["descriptors"][0][None]}


@pytest.mark.parametrize("name", ["c3_pair_s59.npz", "c3_pair_s55.npz", "c5_pair_s19.npz"])
def test_full_size_matches_bit_exact_vs_reference_golden(name):
    """C3 (N=1024, d=128, 30 iters) and C5 (N=2048, d=256, 100 iters): oracle keypoints in,
    match indices out identical to the reference's."""
    g = util.golden(name)
    H, W, d, K, seed = (int(g[k]) for k in ("H", "W", "d", "K", "seed"))
    data = _oracle_pair_inputs(seed, H, W, d, K)
    assert np.array_equal(data["keypoints0"][0].numpy(), g["keypoints0"])     # the oracle is the reference here
    eng, L = _engine(d)
    sd = util.sg_sd(d)
    eng.load_state_dict(L.NET_SUPERGLUE, sd)
    eng.set_debug(True)
    t = {k: v.cuda() for k, v in data.items()}
    m0, m1, ms0, ms1 = _run(eng, t, (1, 1, H, W))
    S = eng.fetch("scores_in")[0]
    an, tag = util.golden("fp64_anchor.npz"), name[:-4]
    util.assert_fp64_anchored(S[:K:8, :K:8], g["scores_in_sub"], an[tag + "/scores_in_sub_f64"], f"{tag} scores_in (every 8th)")
    Z = util.transport_Z(S, eng.fetch("u")[0], eng.fetch("v")[0], K, K, float(sd["bin_score"]))
    util.assert_fp64_anchored(Z[::8, ::8], g["Z_sub"], an[tag + "/Z_sub_f64"], f"{tag} Z (every 8th)")
    nbad0, nbad1 = int((m0 != g["matches0"]).sum()), int((m1 != g["matches1"]).sum())
    assert nbad0 == 0 and nbad1 == 0, f"{nbad0}/{nbad1} match indices differ (fixture decision margin {float(g['margin_decision_gap']):.2e})"
    util.assert_close(ms0, g["matching_scores0"], "matching_scores0")
    util.assert_close(ms1, g["matching_scores1"], "matching_scores1")
    # structural properties of the output
    i = np.nonzero(m0[0] > -1)[0]
    assert np.array_equal(m1[0][m0[0][i]], i), "matches must be mutual"
    assert (ms0[0][i] > eng.cfg.match_threshold).all()


def test_batched_pairs_equal_single_pair_runs():
    d, K, H, W = 128, 1024, 480, 640
    eng, L = _engine(d)
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(d))
    a, b = _oracle_pair_inputs(59, H, W, d, K), _oracle_pair_inputs(55, H, W, d, K)
    both = {k: torch.cat([a[k], b[k]]).cuda() for k in KEYS}
    mb = _run(eng, both, (1, 1, H, W))
    ma = _run(eng, {k: v.cuda() for k, v in a.items()}, (1, 1, H, W))
    mbb = _run(eng, {k: v.cuda() for k, v in b.items()}, (1, 1, H, W))
    for i in range(4):
        assert np.array_equal(mb[i][0], ma[i][0]) and np.array_equal(mb[i][1], mbb[i][0]), \
            "batched pairs must be bit-identical to single-pair runs"


def test_bad_option_is_refused_and_forms_are_reported():
    """imx_set_option refuses unknown keys / values (nothing silently falls back), and imx_timing_form names the form that ran."""
    from image_matching_amd.engine import ImxError
    eng, L = _engine()
    with pytest.raises(ImxError):
        eng.set_option("mfma", "fp16")
    with pytest.raises(ImxError):
        eng.set_option("no_such_option", "1")
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(128))
    g = util.golden("sg_small.npz")
    t = {k: torch.from_numpy(g[k]).cuda() for k in KEYS}
    seen = {}
    for forms, mfma in (("off", "x3"), ("off", "f32"), ("on", "x3")):
        eng.set_option("latency_forms", forms).set_option("mfma", mfma)
        eng.timing_reset()
        eng.set_timing(True)
        _run(eng, t, (1, 1, 120, 160))
        seen[(forms, mfma)] = {r[0]: r[3] for r in eng.timing_report(forms=True)}
        eng.set_timing(False)
    assert seen[("off", "x3")]["qkv_proj"] == "gemm_x3:bf16x3" and seen[("off", "x3")]["attention"] == "attention_h2:f16x2"
    assert seen[("off", "x3")]["gnn_tail"] == "gnn_tail_h2:f16x2" and "gnn_mlp1" not in seen[("off", "x3")]      # round 4: one launch per layer tail
    assert seen[("off", "f32")]["gnn_mlp1"] == "gemm_tiled:f32" and "gnn_tail" not in seen[("off", "f32")]
    assert seen[("off", "f32")]["qkv_proj"] == "gemm_tiled:f32" and seen[("off", "f32")]["attention"] == "attention:f32"
    assert seen[("on", "x3")]["qkv_proj"] == "gemm_small:f32" and seen[("on", "x3")]["attention"] == "attention_split:f32"


def test_fused_layer_tail_of_the_throughput_path_vs_three_launches():
    """gnn_tail_x3 (round 4: mlp.0' -> ReLU -> mlp.3 + residual -> the next layer's q|k|v / final_proj in ONE launch, the hidden
    activations and x' never leaving the wave's registers) against the three gemm_x3 launches it replaces ("gnn_tail" = "unfused"):
    the reference's matches on the full-size fixture from both, GNN output and score matrix equal to rounding (same six-term bf16
    products, another summation order), on a full-size pair, on ragged small counts (partial last workgroup: 5 waves of 8 with rows)
    and on a batch with per-pair device-side counts."""
    g = util.golden("c3_pair_s59.npz")
    H, W, d, K, seed = (int(g[k]) for k in ("H", "W", "d", "K", "seed"))
    data = {k: v.cuda() for k, v in _oracle_pair_inputs(seed, H, W, d, K).items()}
    eng, L = _engine(d)
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(d))
    eng.set_option("latency_forms", "off")
    eng.set_debug(True)
    taps = {}
    for mode in ("fused", "bf16x3", "unfused"):      # fused = gnn_tail_h2 (three fp16 plane products, round 4), bf16x3 = gnn_tail_x3 (six bf16 ones)
        eng.set_option("gnn_tail", mode)
        assert eng.get_option("gnn_tail") == mode
        eng.timing_reset()
        eng.set_timing(True)
        out = _run(eng, data, (1, 1, H, W))
        forms = {r[0]: r[3] for r in eng.timing_report(forms=True)}
        eng.set_timing(False)
        assert ("gnn_tail" in forms) == (mode != "unfused") and ("gnn_mlp1" in forms) == (mode == "unfused"), forms
        if mode != "unfused":
            assert forms["gnn_tail"] == ("gnn_tail_h2:f16x2" if mode == "fused" else "gnn_tail_x3:bf16x3"), forms
            assert "rows_amax" in forms, forms      # (max |x| of layer 0: gnn_tail_h2's bounds and, since round 6, gemm_h2's scale for layer 0's q|k|v)
        assert np.array_equal(out[0], g["matches0"]) and np.array_equal(out[1], g["matches1"]), f"gnn_tail={mode}"
        taps[mode] = (eng.fetch("x").copy(), eng.fetch("scores_in").copy(), out[2].copy())
    scale = np.abs(taps["unfused"][0]).max()
    for mode in ("fused", "bf16x3"):
        assert np.abs(taps[mode][0] - taps["unfused"][0]).max() <= 2e-5 * scale, f"GNN output: {mode} vs three launches"
        np.testing.assert_allclose(taps[mode][2], taps["unfused"][2], rtol=0, atol=2e-5)
    # ragged: 150 x 97 keypoints (R = 160 + 128 = 288 rows: one workgroup, three waves without rows) and a batch of 3 with counts
    gs = util.golden("sg_small.npz")
    for n0, n1 in ((150, 97), (gs["keypoints0"].shape[1], gs["keypoints1"].shape[1])):
        t = {k: torch.from_numpy(gs[k]).cuda() for k in KEYS}
        t = {k: (v[:, :, :n0 if k.endswith("0") else n1] if k.startswith("desc") else v[:, :n0 if k.endswith("0") else n1]).contiguous() for k, v in t.items()}
        res = {}
        for mode in ("fused", "bf16x3", "unfused"):
            eng.set_option("gnn_tail", mode)
            res[mode] = _run(eng, t, (1, 1, 120, 160))
        for mode in ("fused", "bf16x3"):
            assert np.array_equal(res[mode][0], res["unfused"][0]) and np.array_equal(res[mode][1], res["unfused"][1]), (mode, n0, n1)
            np.testing.assert_allclose(res[mode][2], res["unfused"][2], rtol=0, atol=2e-5)


@pytest.mark.parametrize("tail", ["fused", "unfused"])
def test_two_plane_fp16_attention_vs_six_product_bf16(tail):
    """The throughput attention's default form (round 4: operands as two fp16 planes scaled by a power of two from the (side, pair)'s
    q / k / v maxima, three term products per k-step) against the six-product bf16 form it replaces ("attention" = "bf16x3"): the
    reference's matches on the full-size fixture from both, GNN output and matching scores equal to rounding.  The maxima themselves
    -- written by the fused layer tail's epilogue, or by the projection's where the q|k|v comes from a plain GEMM (layer 0; every layer
    under "gnn_tail" = "unfused"; the separate qkv_amax pass only where a pair's padded rows are not whole 128-row tiles) -- equal the maxima of the valid rows of the last layer's q|k|v exactly."""
    g = util.golden("c3_pair_s59.npz")
    H, W, d, K, seed = (int(g[k]) for k in ("H", "W", "d", "K", "seed"))
    data = {k: v.cuda() for k, v in _oracle_pair_inputs(seed, H, W, d, K).items()}
    eng, L = _engine(d)
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(d))
    eng.set_option("latency_forms", "off").set_option("gnn_tail", tail)
    eng.set_debug(True)
    taps = {}
    for mode in ("f16x2", "bf16x3"):
        eng.set_option("attention", mode)
        assert eng.get_option("attention") == mode
        eng.timing_reset()
        eng.set_timing(True)
        out = _run(eng, data, (1, 1, H, W))
        rows = {r[0]: r for r in eng.timing_report(forms=True)}
        eng.set_timing(False)
        assert rows["attention"][3] == ("attention_h2:f16x2" if mode == "f16x2" else "attention_x3:bf16x3"), rows["attention"]
        # (round 5: at these shapes -- padded counts a multiple of 128 -- the projection's own epilogue writes the maxima: no launch)
        assert "qkv_amax" not in rows, rows.get("qkv_amax")
        assert np.array_equal(out[0], g["matches0"]) and np.array_equal(out[1], g["matches1"]), f"attention={mode}"
        taps[mode] = (eng.fetch("x").copy(), eng.fetch("scores_in").copy(), out[2].copy())
        if mode == "f16x2":
            amax, qkv = eng.fetch("amax"), eng.fetch("qkv")
            Kp = (K + 31) // 32 * 32
            for side, r0 in ((0, 0), (1, Kp)):
                for j, what in enumerate("qkv"):
                    want = np.abs(qkv[r0:r0 + K, j * d:(j + 1) * d]).max()
                    assert amax[side, j] == want, f"max |{what}| of side {side}: {amax[side, j]} vs {want} (gnn_tail={tail})"
    scale = np.abs(taps["bf16x3"][0]).max()
    assert np.abs(taps["f16x2"][0] - taps["bf16x3"][0]).max() <= 2e-5 * scale, "GNN output: two fp16 planes vs three bf16 planes"
    np.testing.assert_allclose(taps["f16x2"][2], taps["bf16x3"][2], rtol=0, atol=2e-5)


def test_attention_key_split_and_throughput_forms_agree(monkeypatch):
    """Single pairs take the key-split attention form (32-query workgroups, four waves splitting the keys), batches the
    throughput form.  Both must give the reference's matches on the full-size fixtures (the default B = 1 runs above use the
    split form; here the throughput form is forced), and their matching scores agree to rounding."""
    g = util.golden("c3_pair_s59.npz")
    H, W, d, K, seed = (int(g[k]) for k in ("H", "W", "d", "K", "seed"))
    data = {k: v.cuda() for k, v in _oracle_pair_inputs(seed, H, W, d, K).items()}
    eng, L = _engine(d)
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(d))
    out = {}
    for forms in ("on", "off"):                      # key-split attention + small-M GEMM vs the throughput forms, same handle
        eng.set_option("latency_forms", forms)
        assert eng.get_option("latency_forms") == forms
        out[forms] = _run(eng, data, (1, 1, H, W))
        assert np.array_equal(out[forms][0], g["matches0"]) and np.array_equal(out[forms][1], g["matches1"]), f"latency_forms={forms}"
    np.testing.assert_allclose(out["on"][2], out["off"][2], rtol=0, atol=2e-5)
    monkeypatch.setenv("IMX_LATENCY_FORMS", "off")   # seeds the option of the handles created below (imx_create reads it once)
    for name in ("c3_pair_s55.npz", "c5_pair_s19.npz"):
        test_full_size_matches_bit_exact_vs_reference_golden(name)


@pytest.mark.parametrize("mfma", ["f32", "x3"])
def test_throughput_forms_on_both_matrix_pipes(mfma, monkeypatch):
    """The throughput forms run their fp32 products either on the fp32 MFMA ("mfma" = "f32": attention_kernel, the tiled GEMM of
    gemm.hip) or as six bf16 term products on the bf16 MFMA (attention_x3, gemm_x3; the default).  Both must reproduce the
    reference's matches on the full-size fixtures (head dims 32 and 64) and pass the float64-anchored checks on scores_in and
    Z -- the split is exact, so the bf16 pipe is held to the same bar as the fp32 one."""
    monkeypatch.setenv("IMX_MFMA", mfma)             # the environment seeds the options of every handle created below
    monkeypatch.setenv("IMX_LATENCY_FORMS", "off")
    for name in ("c3_pair_s59.npz", "c3_pair_s55.npz", "c5_pair_s19.npz"):
        test_full_size_matches_bit_exact_vs_reference_golden(name)
    test_small_dense_and_matches_vs_reference_golden()


@pytest.mark.parametrize("name", ["c3_pair_s59.npz", "c5_pair_s19.npz"])
def test_full_size_transport_marginals(name):
    """Size-independent property at BASELINE sizes: the Sinkhorn loop ends on a `v` update
    (superglue_test.py:144-147), so whatever the iteration count the COLUMN marginals of exp(Z) are exact: every
    keypoint column sums to 1, the dustbin column to M; rows are only approximately normalised."""
    g = util.golden(name)
    H, W, d, K, seed = (int(g[k]) for k in ("H", "W", "d", "K", "seed"))
    data = _oracle_pair_inputs(seed, H, W, d, K)
    eng, L = _engine(d)
    sd = util.sg_sd(d)
    eng.load_state_dict(L.NET_SUPERGLUE, sd)
    eng.set_debug(True)
    _run(eng, {k: v.cuda() for k, v in data.items()}, (1, 1, H, W))
    S = eng.fetch("scores_in")[0, :K, :K].astype(np.float64)
    u, v = eng.fetch("u")[0].astype(np.float64), eng.fetch("v")[0].astype(np.float64)
    Z = np.full((K + 1, K + 1), float(sd["bin_score"]))
    Z[:K, :K] = S
    Z = Z + u[:K + 1, None] + v[None, :K + 1] + np.log(2.0 * K)
    P = np.exp(Z)
    np.testing.assert_allclose(P[:, :K].sum(0), 1.0, rtol=3e-4)
    np.testing.assert_allclose(P[:, K].sum(), float(K), rtol=3e-4)
    assert np.all(P[:K].sum(1) > 0.5) and np.all(P[:K].sum(1) < 1.5)


@pytest.mark.parametrize("d", [64, 128, 256])
def test_fused_small_layer_equals_the_three_launch_form(d):
    """Latency form of the GNN layer tail (gnn_small.hip: mlp.0' -> ReLU -> mlp.3 + residual -> the next layer's q|k|v / final_proj
    in one launch) against the three gemm_small launches it replaces ("latency_forms" = "unfused"): the same MFMA sequence, so
    the score matrix, the potentials and every output must be the same BYTES -- ragged counts, two pairs, all three widths."""
    B, N0, N1 = 2, 300, 171
    g = torch.Generator().manual_seed(31 + d)
    t = {"keypoints0": torch.rand(B, N0, 2, generator=g) * torch.tensor([639.0, 479.0]), "keypoints1": torch.rand(B, N1, 2, generator=g) * torch.tensor([639.0, 479.0]),
         "scores0": torch.rand(B, N0, generator=g), "scores1": torch.rand(B, N1, generator=g),
         "descriptors0": torch.nn.functional.normalize(torch.randn(B, d, N0, generator=g), dim=1),
         "descriptors1": torch.nn.functional.normalize(torch.randn(B, d, N1, generator=g), dim=1)}
    tc = {k: v.cuda() for k, v in t.items()}
    n0 = torch.tensor([N0, 123], dtype=torch.int32).cuda()
    n1 = torch.tensor([77, N1], dtype=torch.int32).cuda()
    eng, L = _engine(d)
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(d))
    eng.set_debug(True)
    eng.set_timing(True)
    res = {}
    for forms in ("on", "unfused"):
        eng.set_option("latency_forms", forms)
        eng.timing_reset()
        out = eng.superglue(tc["keypoints0"], tc["scores0"], tc["descriptors0"], (1, 1, 480, 640),
                            tc["keypoints1"], tc["scores1"], tc["descriptors1"], (1, 1, 480, 640), n0, n1)
        torch.cuda.synchronize()
        names = {r[0] for r in eng.timing_report(forms=True)}
        assert ("gnn_layer" in names) == (forms == "on") and ("gnn_mlp1" in names) == (forms == "unfused"), names
        res[forms] = [o.cpu().numpy() for o in out] + [eng.fetch("scores_in"), eng.fetch("x"), eng.fetch("u")]
    for a, b in zip(res["on"], res["unfused"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("tail", ["auto", "unfused"])
def test_qkv_maxima_from_the_projection_epilogue_equal_the_separate_pass(tail, monkeypatch):
    """Round 5: where a layer's q|k|v is a plain projection (layer 0; every layer under "gnn_tail" = "unfused", which is C5's d = 256
    path), gemm_x3's epilogue writes the (side, pair) maxima the two-plane attention scales by -- no `qkv_amax` launch.  A maximum
    does not depend on the order it is taken in: everything downstream must equal the separate pass ("qkv_amax" = "kernel") bit for
    bit, on a full-size pair and on a batch with ragged per-pair counts (rows past a count must not enter the maxima)."""
    g = util.golden("c3_pair_s59.npz")
    H, W, d, K, seed = (int(g[k]) for k in ("H", "W", "d", "K", "seed"))
    one = {k: v.cuda() for k, v in _oracle_pair_inputs(seed, H, W, d, K).items()}
    three = {k: torch.cat([v, v.flip(-1 if k.startswith("desc") else 1), v], 0).contiguous() for k, v in one.items()}
    n0 = torch.tensor([K, 700, 3], dtype=torch.int32, device="cuda")
    n1 = torch.tensor([900, K, 1], dtype=torch.int32, device="cuda")
    eng, L = _engine(d)
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(d))
    eng.set_option("latency_forms", "off")
    eng.set_option("gnn_tail", tail)
    eng.set_debug(True)
    res = {}
    for how in ("epilogue", "kernel"):
        eng.set_option("qkv_amax", how)
        eng.timing_reset()
        eng.set_timing(True)
        a = _run(eng, one, (1, 1, H, W))
        xa = eng.fetch("x").copy()
        forms = {r[0] for r in eng.timing_report(forms=True)}
        eng.set_timing(False)
        assert ("qkv_amax" in forms) == (how == "kernel"), (how, forms)
        b = _run(eng, three, (1, 1, H, W), n0, n1)
        res[how] = (a, xa, b, eng.fetch("x").copy())
    for i in range(3):
        assert np.array_equal(res["epilogue"][0][i], res["kernel"][0][i]) and np.array_equal(res["epilogue"][2][i], res["kernel"][2][i]), i
    assert np.array_equal(res["epilogue"][1], res["kernel"][1]) and np.array_equal(res["epilogue"][3], res["kernel"][3])
    assert np.array_equal(res["epilogue"][0][0], g["matches0"])


@pytest.mark.parametrize("name", ["c3_pair_s59.npz", "c5_pair_s19.npz"])
def test_sinkhorn_grouped_slabs_equal_a_partial_per_slab(name, monkeypatch):
    """Round 5: a Sinkhorn workgroup walks two consecutive 8-row slabs and merges their column partials in registers (half the
    partial traffic).  The merge is the same log-sum-exp with one more pairwise step: potentials equal to 2e-6, match indices equal,
    against the one-partial-per-slab form ("sinkhorn_group" = 1) and against a group of four."""
    g = util.golden(name)
    H, W, d, K, seed = (int(g[k]) for k in ("H", "W", "d", "K", "seed"))
    data = {k: v.cuda() for k, v in _oracle_pair_inputs(seed, H, W, d, K).items()}
    eng, L = _engine(d)
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(d))
    eng.set_debug(True)
    res = {}
    for G in ("1", "2", "4"):
        eng.set_option("sinkhorn_group", G)
        out = _run(eng, data, (1, 1, H, W))
        res[G] = (out, eng.fetch("u").copy(), eng.fetch("v").copy())
    for G in ("2", "4"):
        assert np.array_equal(res[G][0][0], res["1"][0][0]) and np.array_equal(res[G][0][1], res["1"][0][1]), G
        assert np.abs(res[G][1] - res["1"][1]).max() <= 2e-6 * max(1.0, np.abs(res["1"][1]).max()), (G, np.abs(res[G][1] - res["1"][1]).max())
        assert np.abs(res[G][2] - res["1"][2]).max() <= 2e-6 * max(1.0, np.abs(res["1"][2]).max()), (G, np.abs(res[G][2] - res["1"][2]).max())
    assert np.array_equal(res["2"][0][0], g["matches0"])


@pytest.mark.parametrize("name", ["c3_pair_s59.npz", "c5_pair_s19.npz"])
def test_sinkhorn_next_slab_prefetch_is_bit_identical(name, monkeypatch):
    """Round 5: a grouped Sinkhorn workgroup touches the lines of its next slab's rows before the current slab's column pass
    ("sinkhorn_prefetch" = "on").  Only the timing of the loads changes: potentials and matches bit for bit those of the form
    without it, for both group sizes."""
    g = util.golden(name)
    H, W, d, K, seed = (int(g[k]) for k in ("H", "W", "d", "K", "seed"))
    data = {k: v.cuda() for k, v in _oracle_pair_inputs(seed, H, W, d, K).items()}
    eng, L = _engine(d)
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(d))
    eng.set_debug(True)
    for G in ("2", "4"):
        eng.set_option("sinkhorn_group", G)
        res = {}
        for pf in ("0", "1"):
            eng.set_option("sinkhorn_prefetch", pf)
            out = _run(eng, data, (1, 1, H, W))
            res[pf] = (out, eng.fetch("u").copy(), eng.fetch("v").copy())
        for i in range(3):
            assert np.array_equal(res["0"][0][i], res["1"][0][i]), (G, i)
        assert np.array_equal(res["0"][1], res["1"][1]) and np.array_equal(res["0"][2], res["1"][2]), G


@pytest.mark.parametrize("name", ["c3_pair_s59.npz"])
def test_two_query_blocks_per_wave_are_bit_identical(name):
    """Round 6: attention_h2q2_kernel (two 32-query blocks per wave, the Q fragments of both in wave-private LDS) runs the one-block
    kernel's instructions per query in the same order: GNN output, scores and matches bit for bit, on a full-size pair ("2" forces it
    at one pair; "auto" takes it from 16 pairs up) and on a batch of three with ragged counts (keys and queries past a count)."""
    g = util.golden(name)
    H, W, d, K, seed = (int(g[k]) for k in ("H", "W", "d", "K", "seed"))
    one = {k: v.cuda() for k, v in _oracle_pair_inputs(seed, H, W, d, K).items()}
    three = {k: torch.cat([v, v.flip(-1 if k.startswith("desc") else 1), v], 0).contiguous() for k, v in one.items()}
    n0 = torch.tensor([K, 700, 3], dtype=torch.int32, device="cuda")
    n1 = torch.tensor([900, K, 1], dtype=torch.int32, device="cuda")
    eng, L = _engine(d)
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(d))
    eng.set_option("latency_forms", "off")
    eng.set_debug(True)
    res = {}
    for qb in ("1", "2"):
        eng.set_option("attention_qblocks", qb)
        assert eng.get_option("attention_qblocks") == qb
        a = _run(eng, one, (1, 1, H, W))
        xa, sa = eng.fetch("x").copy(), eng.fetch("scores_in").copy()
        b = _run(eng, three, (1, 1, H, W), n0, n1)
        res[qb] = (a, xa, sa, b, eng.fetch("x").copy())
    for i in range(4):
        assert np.array_equal(res["1"][0][i], res["2"][0][i]) and np.array_equal(res["1"][3][i], res["2"][3][i]), i
    assert np.array_equal(res["1"][1], res["2"][1]) and np.array_equal(res["1"][2], res["2"][2]) and np.array_equal(res["1"][4], res["2"][4])
    assert np.array_equal(res["2"][0][0], g["matches0"])


@pytest.mark.parametrize("name", ["c3_pair_s59.npz", "c5_pair_s19.npz"])
def test_sinkhorn_fused_merge_equals_the_merge_kernel(name):
    """Round 6 (one launch per Sinkhorn iteration): the last-arriving slab workgroups of a pair merge its column
    partials themselves ("sinkhorn_merge" = fused, the default) instead of a second launch (sinkhorn_vmerge, "kernel").  Both run the
    same merge routine on the same 16 chains per column: potentials and matches must agree bit for bit -- on a full pair, for every
    group size (1 = more groups than column blocks: most of a pair's workgroups leave without merging; 4 at one pair = fewer groups
    than blocks: a merger takes several), on a batch of three with ragged counts (a pair with one column: a single block; three rows:
    a single group), and the word that records a merger giving up its wait must stay zero."""
    g = util.golden(name)
    H, W, d, K, seed = (int(g[k]) for k in ("H", "W", "d", "K", "seed"))
    one = {k: v.cuda() for k, v in _oracle_pair_inputs(seed, H, W, d, K).items()}
    three = {k: torch.cat([v, v.flip(-1 if k.startswith("desc") else 1), v], 0).contiguous() for k, v in one.items()}
    n0 = torch.tensor([K, 700, 3], dtype=torch.int32, device="cuda")
    n1 = torch.tensor([900, K, 1], dtype=torch.int32, device="cuda")
    eng, L = _engine(d)
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(d))
    eng.set_option("latency_forms", "off")
    eng.set_debug(True)
    for G in ("auto", "1", "2", "4"):
        eng.set_option("sinkhorn_group", G)
        res = {}
        for how in ("kernel", "fused"):
            eng.set_option("sinkhorn_merge", how)
            assert eng.get_option("sinkhorn_merge") == how
            a = _run(eng, one, (1, 1, H, W))
            ua, va = eng.fetch("u").copy(), eng.fetch("v").copy()
            if how == "fused":
                assert int(eng.fetch("sk_merge_cnt").view(np.uint32)[-1]) == 0, "a merging workgroup gave up waiting"
            b = _run(eng, three, (1, 1, H, W), n0, n1)
            res[how] = (a, ua, va, b, eng.fetch("u").copy(), eng.fetch("v").copy())
            if how == "fused":
                assert int(eng.fetch("sk_merge_cnt").view(np.uint32)[-1]) == 0, "a merging workgroup gave up waiting (batch of three)"
        for i in range(4):
            assert np.array_equal(res["kernel"][0][i], res["fused"][0][i]) and np.array_equal(res["kernel"][3][i], res["fused"][3][i]), (G, i)
        for i in (1, 2, 4, 5):
            assert np.array_equal(res["kernel"][i], res["fused"][i]), (G, i, np.abs(res["kernel"][i] - res["fused"][i]).max())
    assert np.array_equal(res["fused"][0][0], g["matches0"])


def test_sinkhorn_fused_merge_on_a_small_pair():
    """The fused merge where the slab tile is smaller than the merge's 8 KB of chains (the launcher raises the dynamic LDS) and a pair
    has fewer groups than workgroup slots by far: sg_small's pair (a few hundred keypoints), throughput forms, every group size."""
    g = util.golden("sg_small.npz")
    eng, L = _engine()
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(128))
    eng.set_option("latency_forms", "off")
    eng.set_debug(True)
    t = {k: torch.from_numpy(g[k]).cuda() for k in KEYS}
    for G in ("auto", "1", "2", "4"):
        eng.set_option("sinkhorn_group", G)
        res = {}
        for how in ("kernel", "fused"):
            eng.set_option("sinkhorn_merge", how)
            out = _run(eng, t, (1, 1, 120, 160))
            res[how] = (out, eng.fetch("u").copy(), eng.fetch("v").copy())
        assert int(eng.fetch("sk_merge_cnt").view(np.uint32)[-1]) == 0
        for i in range(4):
            assert np.array_equal(res["kernel"][0][i], res["fused"][0][i]), (G, i)
        assert np.array_equal(res["kernel"][1], res["fused"][1]) and np.array_equal(res["kernel"][2], res["fused"][2]), G
    assert np.array_equal(res["fused"][0][0], g["matches0"])

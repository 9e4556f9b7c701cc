"""Round 4 (VERDICT r3 task 1): the north_star bar, literally -- |hip - ref| <= 1e-4 + 1e-4|ref| ELEMENT-WISE on gnn17,
scores_in and Z and ZERO differing match indices -- on the "t" SuperGlue weight set (synth.SGT_GAINS: scores_in std ~ 5,
bin_score = mean + 2 sigma; the reference's own fp32 forward is inside the same bar of its float64 self there, so the bar is
attainable) and on UNSELECTED seeds (C3 1000-1031, C5 2000-2007; tests/golden/make_golden.py --strict-set imports the
reference, nothing is rejected).  No envelope, floor or error-scaled margin anywhere in this file.  ONE rule comes from the bar
itself: where the reference's candidate matching score is within 1e-4 of match_threshold (util.threshold_band_rows: a property
of the reference's output alone, ~0.5 rows per pair), a score that agrees to 1e-4 can sit on either side of `mscores0 >
match_threshold` (superglue_test.py:281), so the pair may come back matched or unmatched -- with the reference's candidate index;
every such row is counted and printed.  Needs an MI355X.

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
    from tests import oracle_jobs
    g = util.golden(name)
    per_seed = []
    for s, (data, ref) in enumerate(oracle_jobs.pool_map(oracle_jobs.strict_inputs_job, [(name, s) for s in range(len(g["seeds"]))])):
        for key, (mine, fx) in util.strict_samples(g, s, ref["gnn0"], ref["gnn1"], ref["scores_in"], ref["Z"]).items():
            util.assert_close(mine, fx, f"oracle vs the reference's {key} ({name} seed {int(g['seeds'][s])})", atol=1e-5, rtol=1e-5)
        per_seed.append(({k: torch.from_numpy(v) for k, v in data.items()}, ref))
    _INPUTS[name] = (g, per_seed)
    return _INPUTS[name]


def _pair_taps(eng, K):
    """A single pair's GNN output out of the library's tap, laid out like the reference's: gnn0 / gnn1 (d, N)."""
    Kp = (K + 31) // 32 * 32
    x = eng.fetch("x")
    return x[:K].T, x[Kp:Kp + K].T


@pytest.mark.parametrize("forms,mfma,attention,linear", [("auto", "x3", "auto", "auto"), ("off", "x3", "auto", "auto"), ("off", "x3", "auto", "bf16x3"),
                                                        ("off", "x3", "bf16x3", "auto"), ("off", "f32", "auto", "auto")])
@pytest.mark.parametrize("name", ["strict_c3.npz", "strict_c5.npz"])
def test_strict_bar_superglue_every_form(name, forms, mfma, attention, linear):
    """SuperGlue alone, the reference's keypoints injected, on every kernel form a caller can reach (latency forms; the
    throughput forms bench.py times -- attention and the plain linear layers (gemm_h2: every layer's q|k|v, mlp.0', mlp.3 at d = 256,
    layer 0's q|k|v at d = 128) on two fp16 planes; the same with the linear layers on three bf16 planes ("linear" = bf16x3); with the
    attention on three bf16 planes (which takes the linear layers there too); their fp32-MFMA counterparts): gnn17, scores_in and Z element-wise inside
    1e-4 + 1e-4|ref| of the oracle's (every element) and of the reference's (fixture samples); matches0 / matches1 equal to the
    reference's; matching scores at the same tolerance."""
    g, per_seed = _strict_inputs(name)
    H, W, d, K = (int(g[k]) for k in ("H", "W", "d", "K"))
    from image_matching_amd import _lib as L
    from image_matching_amd.engine import Engine
    eng = Engine(util.sp_config(d, K), util.sg_config(d), "cuda")
    sd_sg = util.sg_sd(d, variant="t")
    eng.load_state_dict(L.NET_SUPERGLUE, sd_sg)
    eng.set_option("latency_forms", forms).set_option("mfma", mfma).set_option("attention", attention).set_option("linear", linear)
    eng.set_debug(True)
    alpha = float(sd_sg["bin_score"])
    want_h2 = forms == "off" and mfma == "x3" and attention == "auto" and linear == "auto"
    worst = {"gnn17": 0.0, "scores_in": 0.0, "Z": 0.0, "mscores": 0.0}
    thr, band, other = float(util.sg_config(d)["match_threshold"]), 0, 0
    for s, seed in enumerate(g["seeds"]):
        data, ref = per_seed[s]
        if s == 0:
            eng.timing_reset()
            eng.set_timing(True)
        out = eng.superglue(data["keypoints0"].cuda(), data["scores0"].cuda(), data["descriptors0"].cuda(), (1, 1, H, W),
                            data["keypoints1"].cuda(), data["scores1"].cuda(), data["descriptors1"].cuda(), (1, 1, H, W))
        torch.cuda.synchronize()
        if s == 0:                        # the forms that ran are the forms the parameters name
            ran = {r[0]: r[3] for r in eng.timing_report(forms=True)}
            eng.set_timing(False)
            if forms == "off" and mfma == "x3":
                assert ran["qkv_proj"] == ("gemm_h2:f16x2" if want_h2 else "gemm_x3:bf16x3"), ran
                if d == 256:
                    assert ran["gnn_mlp1"] == ran["gnn_mlp2"] == ran["final_proj"] == ("gemm_h2:f16x2" if want_h2 else "gemm_x3:bf16x3"), ran
        m0, m1, ms0, ms1 = (o.cpu().numpy() for o in out)
        g0, g1 = _pair_taps(eng, K)
        S = eng.fetch("scores_in")[0, :K, :K]
        Z = util.transport_Z(S, eng.fetch("u")[0], eng.fetch("v")[0], K, K, alpha)
        tag = f"{name} seed {seed} [{forms}/{mfma}/{attention}/linear={linear}]"
        for key, mine, full in (("gnn17", np.stack([g0, g1]), np.stack([ref["gnn0"], ref["gnn1"]])), ("scores_in", S, ref["scores_in"]), ("Z", Z, ref["Z"])):
            util.assert_close(mine, full, f"{tag}: {key} vs the oracle, every element")
            worst[key] = max(worst[key], util.tolerance_used(mine, full))
        for key, (mine, fx) in util.strict_samples(g, s, g0, g1, S, Z).items():
            util.assert_close(mine, fx, f"{tag}: {key} vs the reference's sample")
        other += util.strict_index_check(g, s, m0[0], m1[0], thr, tag)
        band += len(util.threshold_band_rows(g, s, thr)[0])
        same0, same1 = m0[0] == g["matches0"][s], m1[0] == g["matches1"][s]
        util.assert_close(ms0[0][same0], g["mscores0"][s][same0], f"{tag}: matching_scores0")
        util.assert_close(ms1[0][same1], g["mscores1"][s][same1], f"{tag}: matching_scores1")
        worst["mscores"] = max(worst["mscores"], util.tolerance_used(ms0[0][same0], g["mscores0"][s][same0]))
    n = len(g["seeds"])
    print(f"[strict] {name} [{forms}/{mfma}/attention={attention}/linear={linear}]: {n} unselected seeds, {2 * K * n} match indices: 0 differ outside the threshold band; {band} rows have their "
          f"reference score within 1e-4 of the threshold, {other} of them sit on the other side here; worst fraction of the 1e-4+1e-4|ref| tolerance used: "
          + ", ".join(f"{k} {v:.3f}" for k, v in worst.items()))


def test_first_layers_with_undamped_gains_on_the_throughput_forms():
    """the 't' weight set damps the residual branch (mlp.3 x 0.1, q / k x 0.5), so an error of the two-plane attention or
    of the fused fp16 layer tail reaches gnn17 / scores_in / Z attenuated layer after layer -- the strict bar above is least sensitive
    exactly where round 4's new kernels sit.  Here the DEFAULT gains (no damping: attention logits four times larger, the residual
    branch at full size) on the same kernels (latency forms off: attention_h2, gnn_tail_h2) at C3 size, and the outputs of the first
    two layers -- before any attenuation can accumulate -- element-wise at 1e-4 + 1e-4|ref| against the oracle; the keypoint encoder
    too.  (After 18 undamped layers the reference's own fp32 result is 2e-4..2.6e-3 from float64: that end is float64-anchored in
    tests/test_gpu_parity_r2.py.)"""
    from oracle import superglue_ref
    from image_matching_amd import _lib as L
    from image_matching_amd.engine import Engine
    g, per_seed = _strict_inputs("strict_c3.npz")
    H, W, d, K = (int(g[k]) for k in ("H", "W", "d", "K"))
    eng = Engine(util.sp_config(d, K), util.sg_config(d), "cuda")
    sd = util.sg_sd(d)                                   # default gains
    eng.load_state_dict(L.NET_SUPERGLUE, sd)
    eng.set_option("latency_forms", "off")
    eng.set_debug(True)
    Kp = (K + 31) // 32 * 32
    worst = {}
    for s in range(2):
        data, _ = per_seed[s]
        eng.timing_reset()
        eng.set_timing(True)
        eng.superglue(data["keypoints0"].cuda(), data["scores0"].cuda(), data["descriptors0"].cuda(), (1, 1, H, W),
                      data["keypoints1"].cuda(), data["scores1"].cuda(), data["descriptors1"].cuda(), (1, 1, H, W))
        torch.cuda.synchronize()
        forms = {r[0]: r[3] for r in eng.timing_report(forms=True)}
        eng.set_timing(False)
        assert forms["attention"] == "attention_h2:f16x2" and forms["gnn_tail"] == "gnn_tail_h2:f16x2", forms
        full = dict(data, image_shape0=(1, 1, H, W), image_shape1=(1, 1, H, W))
        dn = superglue_ref.superglue_forward(full, sd, util.sg_config(d), return_dense=True)["dense"]
        for tap, (r0, r1) in (("kenc", (dn["kenc0"], dn["kenc1"])), ("gnn0", dn["gnn_taps"][0]), ("gnn1", dn["gnn_taps"][1])):
            a = eng.fetch(tap)
            for side, mine, ref in ((0, a[:K].T, r0[0].numpy()), (1, a[Kp:Kp + K].T, r1[0].numpy())):
                util.assert_close(mine, ref, f"default gains, seed {int(g['seeds'][s])}: {tap} side {side}, every element")
                worst[tap] = max(worst.get(tap, 0.0), util.tolerance_used(mine, ref))
    print("[strict] undamped gains, throughput forms, first layers: worst fraction of the tolerance used: " + ", ".join(f"{k} {v:.3f}" for k, v in worst.items()))


def _matching_t(d, K):
    from image_matching_amd.superglue.models.matching_test import Matching
    m = Matching({"superpoint": util.sp_config(d, K), "superglue": util.sg_config(d)}).eval().to("cuda")
    m.superpoint.load_state_dict(util.sp_sd(d))
    m.superglue.load_state_dict(util.sg_sd(d, variant="t"))
    return m


@pytest.mark.parametrize("name,B,first", [("strict_c3.npz", 64, 0), ("strict_c5.npz", 8, 0), ("strict_c5.npz", 8, 8)])
def test_strict_bar_as_one_batched_call_from_images(name, B, first):
    """The whole HIP path, images in, as ONE imx_match_pairs call of B pairs -- the call, batch size and kernel forms bench.py
    times (asserted) -- on the strict set: keypoint sets identical to the reference's, zero differing matches (threshold-band rule of
    the module docstring), matching scores at 1e-4 + 1e-4|ref|.  Dense tensors, two statements: (i) stage by stage at the north_star
    bar, every element -- the call's SuperPoint outputs against the oracle's SuperPoint, and the call's gnn17 / scores_in / Z against
    the oracle's SuperGlue run on those same SuperPoint outputs; (ii) images in, against the reference's samples: SuperPoint's
    within-tolerance differences (descriptors ~4e-6) are amplified by the GNN.  Round 6: that amplification is MEASURED per element
    -- the oracle's SuperGlue in float64 on the call's own SuperPoint outputs and on the reference's (oracle_jobs.stage_job) -- and
    gnn17 / scores_in / Z of the call are held to 1x of 1e-4 + 1e-4|f64| against the float64 evaluation on the call's own inputs (no
    envelope term); every sample outside 1x of the reference's fp32 value is printed with the measured input response at that element.
    `first`: the call's pairs are the fixture's seeds first .. first + B - 1 (strict_c5's sixteen seeds go through two 8-pair calls)."""
    g = util.golden(name)
    H, W, d, K = (int(g[k]) for k in ("H", "W", "d", "K"))
    n = len(g["seeds"])
    m = _matching_t(d, K)
    sidx = [(first + b) % n for b in range(B)]        # fixture seed index of pair b
    ims = {s: util.pair(int(g["seeds"][s]), H, W) for s in set(sidx)}
    i0 = torch.cat([ims[s][0] for s in sidx]).cuda()
    i1 = torch.cat([ims[s][1] for s in sidx]).cuda()
    eng = m._shared.get_engine([0, 1])
    eng.timing_reset()
    eng.set_timing(True)
    eng.set_debug(True)
    out = m.match_batch(i0, i1, want_desc=True)
    torch.cuda.synchronize()
    forms = {r[0]: r[3] for r in eng.timing_report(forms=True)}
    eng.set_timing(False)
    assert forms["qkv_proj"] == "gemm_h2:f16x2" and forms["attention"] == "attention_h2:f16x2" and forms["conv2a"] == "conv3x3_wino24p:f16x2", forms
    alpha = float(util.sg_sd(d, variant="t")["bin_score"])
    # the SuperGlue STAGE inside this very call, at the north_star bar: every distinct pair's taps against the oracle's SuperGlue run on
    # the library's OWN SuperPoint outputs of the call (identical inputs on both sides), every element; those SuperPoint outputs
    # against the oracle's SuperPoint (keypoints matched by coordinate: near-tied scores swap neighbours in the top-k order); and the
    # float64 runs behind the images-in statement (the oracle in a process pool)
    from tests import oracle_jobs
    X, S, U, V = eng.fetch("x"), eng.fetch("scores_in"), eng.fetch("u"), eng.fetch("v")
    Kp = (K + 31) // 32 * 32
    worst = {"gnn17": 0.0, "scores_in": 0.0, "Z": 0.0, "sp_scores": 0.0, "sp_descriptors": 0.0}
    n_stage = min(B, n)            # every distinct pair of the call
    owns = []
    for b in range(n_stage):
        own = {}
        for side in ("0", "1"):
            kp, sc, ds = (out[k + side][b].cpu() for k in ("keypoints", "scores", "descriptors"))
            own["keypoints" + side], own["scores" + side], own["descriptors" + side] = kp.numpy(), sc.numpy(), np.ascontiguousarray(ds.t().numpy())
        owns.append(own)
    strides = (int(g["stride_s"]), int(g["stride_g"]))
    res3 = oracle_jobs.pool_map(oracle_jobs.stage_job, [(d, K, H, W, int(g["seeds"][sidx[b]]), owns[b], strides) for b in range(n_stage)])
    res = [(r[0], r[1]) for r in res3]
    measured = [res3[b][2] if b < n_stage else None for b in range(B)]
    summary = util.strict_compare_batch(g, out, eng, B, alpha, float(util.sg_config(d)["match_threshold"]), seed_idx=sidx, measured=measured)
    print(f"[strict e2e] {name} seeds {first}..{first + B - 1} as one call of {B} pairs: {summary}")
    for b, (sp, dn) in enumerate(res):
        for si, side in enumerate(("0", "1")):
            o, kp = sp[si], owns[b]["keypoints" + side]
            pos = {tuple(p): i for i, p in enumerate(o["keypoints"].astype(int))}
            perm = np.array([pos[tuple(p)] for p in kp.astype(int)])
            util.assert_close(owns[b]["scores" + side], o["scores"][perm], f"pair {b} side {side}: keypoint scores vs the oracle's SuperPoint")
            util.assert_close(owns[b]["descriptors" + side], o["descriptors"][:, perm], f"pair {b} side {side}: descriptors vs the oracle's SuperPoint")
            worst["sp_scores"] = max(worst["sp_scores"], util.tolerance_used(owns[b]["scores" + side], o["scores"][perm]))
            worst["sp_descriptors"] = max(worst["sp_descriptors"], util.tolerance_used(owns[b]["descriptors" + side], o["descriptors"][:, perm]))
        g0, g1 = X[b * Kp:b * Kp + K].T, X[B * Kp + b * Kp:B * Kp + b * Kp + K].T
        Z = util.transport_Z(S[b], U[b], V[b], K, K, alpha)
        for key, mine, ref in (("gnn17", np.stack([g0, g1]), np.stack([dn["gnn0"], dn["gnn1"]])), ("scores_in", S[b, :K, :K], dn["scores_in"]), ("Z", Z, dn["Z"])):
            util.assert_close(mine, ref, f"pair {b} of the {B}-pair call: {key} vs the oracle's SuperGlue on the same inputs, every element")
            worst[key] = max(worst[key], util.tolerance_used(mine, ref))
    print(f"[strict e2e] {name}: per-stage parity inside the {B}-pair call (all {n_stage} distinct pairs, every element), worst fraction of the tolerance used: "
          + ", ".join(f"{k} {v:.3f}" for k, v in worst.items()))
    if summary.get("measured_conditioning"):
        print(f"[strict e2e] {name} seeds {first}..{first + B - 1}: images in, against the float64 SuperGlue on the call's OWN SuperPoint outputs (1x bar, every fixture "
              f"sample), and the measured response of that float64 SuperGlue to the SuperPoint differences (delta, in tolerances of the reference's value): {summary['measured_conditioning']}")
    if summary.get("outliers"):
        print(f"[strict e2e] {name}: samples outside 1x of the tolerance against the REFERENCE's fp32 value (images in), each with the measured input response there: "
              + "; ".join(f"pair {o['pair']} {o['tensor']} hip-ref {o['hip_vs_ref_in_tolerances']}x = input response {o['measured_input_response_in_tolerances']}x "
                          f"(+ hip vs f64 on its own inputs {o['hip_vs_f64_on_its_own_inputs_in_tolerances']}x)" for o in summary["outliers"][:16]))
    m0 = out["matches0"].cpu().numpy()
    for b in range(n, B):
        assert np.array_equal(m0[b], m0[b - n]), f"pair {b} differs from its copy at {b - n}"

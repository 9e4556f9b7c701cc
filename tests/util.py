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


def sg_sd(d=128, kenc=None, n_layers=18, variant="default"):
    return to_torch(synth.make_superglue_state_dict(d, kenc, n_layers, variant=variant))


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


def assert_close(a, b, what, atol=ATOL, rtol=RTOL):
    """|a-b| <= atol + rtol*|b| element-wise -- the north_star tolerance, never relaxed by the tensor scale."""
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a.astype(np.float64) - b.astype(np.float64))
    tol = atol + rtol * np.abs(b.astype(np.float64))
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} elements out of tolerance; max err {err.max():.3e} "
                           f"(at |ref| {np.abs(b).reshape(-1)[err.argmax()]:.3e}), max |ref| {np.abs(b).max():.3e}")


def sinkhorn_drift_bound(u, v, iters):
    """What ANY fp32 evaluation of the log-domain Sinkhorn loop may be away from exact arithmetic on entries of Z, whatever its
    operation order (round 3, tools/debug_sinkhorn.py).  A row i whose mass sits on one column j makes the pair (u_i, v_j) a
    marginally stable direction of the iteration: u_i <- c - RN(S_ij + v_j), v_j <- c' - (S_ij + u_i) (the second subtraction is
    exact), so the rounding residual of RN(S_ij + v_j) -- the SAME value every iteration once the loop has settled -- is re-added
    each time: u_i drifts linearly, by up to half a spacing of |u_i| per iteration, v_j by the opposite amount.  Measured on the
    GPU: 1.6e-5 per iteration at |u| = 277 (spacing 3.05e-5) over 20 iterations; the reference's own fp32 loop does the same with
    another residual (its fp32-vs-float64 envelope reaches 2.6e-3 on sweep seed 1002).  Z_ij = S_ij + u_i + v_j is untouched on the
    pair itself (the two drifts cancel) and moves on the rest of row i / column j -- entries whose exp(Z) is negligible; exp(Z), what
    the reference consumes (superglue_test.py:280), is not affected.  Bound per entry of Z: iters x spacing(max |u|, |v|)."""
    top = max(float(np.abs(np.asarray(u, np.float64)).max()), float(np.abs(np.asarray(v, np.float64)).max()))
    return iters * float(np.spacing(np.float32(top)))


def _ratio(a, b):
    """a / b for the diagnostic prints; a zero envelope (a 1x1 pair: the oracle's evaluations can agree with float64 to the bit) prints inf."""
    return float(a) / float(b) if b else float("inf") if a else 1.0


def additive_part(e):
    """The part of an error field e(i, j) of Z that is a row constant plus a column constant, e_u(i) + e_v(j) -- what an error of the
    Sinkhorn potentials u, v looks like in Z = S + u + v - norm -- by two-way means (least squares)."""
    e = np.asarray(e, np.float64)
    return e.mean(1, keepdims=True) + e.mean(0, keepdims=True) - e.mean()


def assert_sinkhorn_anchored(Z, Z32s, Z64, what, drift_floor, iters=None, c=2.0, c_max=2.5):
    """Sinkhorn alone on a given score matrix (round 4, VERDICT r3 task 7: the floor must not be able to mask a regression).  The
    library's Z against the float64 optimal transport, with the error split in two (tools/sinkhorn_growth_diag.py measured both):
      * the NON-additive part (everything that is not e_u(i) + e_v(j): the exponentials, the adds that assemble Z) must be as close to
        float64 as the oracle's fp32 evaluations are -- rms within c, max within c_max of their envelope, NO floor;
      * the additive part is an error of the potentials.  Potentials of (nearly) decoupled blocks of the plan are marginally stable
        directions of the fp32 iteration and drift linearly with the iteration count in ANY fp32 evaluation (measured: equal to the
        oracle's after one iteration, then +~0.3 spacings per iteration, purely additive; the reference's own loop drifts the same way
        on its own residuals): bounded by max(c_max x the oracle's additive envelope, drift_floor = iterations x spacing).
    The callers add a ONE-iteration run held to the plain c / c_max criterion, which is where an error of a log-sum-exp or of the
    slab merge would show (it cannot hide in a drift that has not happened yet)."""
    Z, Z64 = np.asarray(Z, np.float64), np.asarray(Z64, np.float64)
    refs = [np.asarray(r, np.float64) for r in Z32s]
    eh = Z - Z64
    ah, rh = additive_part(eh), eh - additive_part(eh)
    rms = lambda x: float(np.sqrt((x ** 2).mean()))
    ra = [additive_part(r - Z64) for r in refs]
    rr = [(r - Z64) - a for r, a in zip(refs, ra)]
    env_rem_max, env_rem_rms = max(np.abs(x).max() for x in rr), max(rms(x) for x in rr)
    env_add_max = max(np.abs(x).max() for x in ra)
    print(f"[sinkhorn-anchored] {what}: non-additive error max {np.abs(rh).max():.2e} rms {rms(rh):.2e} vs the oracle's fp32 envelope {env_rem_max:.2e} / {env_rem_rms:.2e} "
          f"(x{_ratio(np.abs(rh).max(), env_rem_max):.2f} / x{_ratio(rms(rh), env_rem_rms):.2f}); additive (potential) error max {np.abs(ah).max():.2e} vs the oracle's {env_add_max:.2e}, drift bound {drift_floor:.2e}")
    # one spacing at the potentials' magnitude = ONE rounding of the adds that assemble Z = (S + u) + v - norm: drifted potentials round
    # differently there, so two evaluations legitimately differ by it entry by entry; on a tiny problem (2x17: 54 entries) five draws
    # of the oracle do not sample that maximum, hence the explicit term.  It is 1/iterations of the drift bound -- no hiding place.
    one_rounding = drift_floor / iters if iters else 0.0
    assert rms(rh) <= max(c * env_rem_rms, one_rounding / 3) and np.abs(rh).max() <= max(c_max * env_rem_max, one_rounding), \
        f"{what}: the non-additive part of the error exceeds {c}x (rms) / {c_max}x (max) the oracle's own fp32 envelope: max {np.abs(rh).max():.2e} vs {env_rem_max:.2e}, rms {rms(rh):.2e} vs {env_rem_rms:.2e}"
    assert np.abs(ah).max() <= max(c_max * env_add_max, drift_floor), \
        f"{what}: the potentials are off by {np.abs(ah).max():.2e}: more than {c_max}x the oracle's {env_add_max:.2e} and than the fp32 drift bound {drift_floor:.2e}"


def assert_plan_close(Z_hip, Z_ref, what):
    """exp(Z) -- the transport plan whose maxima become the matching scores -- at the north_star tolerance, element-wise."""
    assert_close(np.exp(np.asarray(Z_hip, np.float64)), np.exp(np.asarray(Z_ref, np.float64)), what + ": exp(Z)")


def assert_plan_anchored(Z_hip, Z32s, Z64, what, c=2.5):
    """exp(Z) against the float64 plan at the north_star tolerance -- unless the oracle's OWN fp32 evaluations of the same problem do
    not reach it (random descriptors drive |S| to several hundred: the drifted potentials move exp(Z) by more than 1e-4 on a few
    entries in any fp32 evaluation): then the limit is c x the worst of theirs.  Same anchoring as the sweeps' `used_P`."""
    P64 = np.exp(np.asarray(Z64, np.float64))
    used = tolerance_used(np.exp(np.asarray(Z_hip, np.float64)), P64)
    own = max(tolerance_used(np.exp(np.asarray(r, np.float64)), P64) for r in Z32s)
    assert used <= max(1.0, c * own), f"{what}: exp(Z) uses {used:.2f} of the 1e-4 + 1e-4|ref| tolerance against float64 (the oracle's fp32 evaluations: {own:.2f})"


def assert_fp64_anchored(hip, ref32, f64, what, c=2.0, c_max=2.5, floor=0.0):
    """For long fp32 reductions (GNN features after 18 layers, the score matrix, the transport matrix Z) two
    correct fp32 evaluation orders differ by more than 1e-4 at small |ref| -- the reference's own fp32 result
    is that far from the exact value.  So both fp32 results are measured against the SAME float64 evaluation of
    the reference module (tests/golden/make_golden.py: sg_dense_f64) and the HIP result must be as close to it
    as the reference's fp32 result is: rms error within a factor c, max error within c_max -- neither scaled by the
    tensor.  (The maximum over ~10^5 samples of a heavy-tailed error is a noisy statistic: the SAME arithmetic under two
    attention tilings measured 1.9x and 2.2x on C5 while the rms moved 1.46x -> 1.63x; the rms is the robust one.)
    `ref32` may be a LIST of fp32 evaluations of the same quantity (e.g. the oracle on permuted inputs): the envelope is then the
    largest of their errors -- one fp32 evaluation of a small problem is a single draw of the rounding noise and can be
    several times luckier than the next.
    `floor`: an absolute error every fp32 evaluation is entitled to (sinkhorn_drift_bound); the limits are max(c x envelope, floor)."""
    as64 = lambda x: np.asarray(x.detach().cpu() if isinstance(x, torch.Tensor) else x, dtype=np.float64)
    refs = [as64(r) for r in (ref32 if isinstance(ref32, (list, tuple)) else [ref32])]
    hip, f64 = as64(hip), as64(f64)
    assert hip.shape == f64.shape == refs[0].shape, f"{what}: shapes {hip.shape} {refs[0].shape} {f64.shape}"
    eh = np.abs(hip - f64)
    mh, rh = eh.max(), np.sqrt((eh ** 2).mean())
    mr = max(np.abs(r - f64).max() for r in refs)
    rr = max(np.sqrt(((r - f64) ** 2).mean()) for r in refs)
    print(f"[fp64-anchored] {what}: max err hip {mh:.3e} vs reference-fp32 {mr:.3e} (x{_ratio(mh, mr):.2f}); "
          f"rms hip {rh:.3e} vs {rr:.3e} (x{_ratio(rh, rr):.2f}); max|f64| {np.abs(f64).max():.1f}; "
          f"outside 1e-4+1e-4|ref|: hip-vs-reference {outside_fraction(hip, refs[0]):.2e}, hip-vs-f64 {outside_fraction(hip, f64):.2e}, "
          f"reference-vs-f64 {outside_fraction(refs[0], f64):.2e}" + (f" (envelope over {len(refs)} fp32 evaluations)" if len(refs) > 1 else ""))
    # the floor is a per-ENTRY worst case (iterations x spacing): it bounds the maximum only -- as an rms limit it would be ~100x the
    # measured rms and make the rms check vacuous
    assert rh <= c * rr and mh <= max(c_max * mr, floor), \
        (f"{what}: HIP is further from the float64 evaluation than {c}x (rms) / {c_max}x (max) the reference's own fp32 result"
         f"{f' and than the fp32 drift bound {floor:.2e}' if floor else ''}: max {mh:.3e} vs {mr:.3e}, rms {rh:.3e} vs {rr:.3e}")
    return _ratio(mh, mr), _ratio(rh, rr)


def sinkhorn_fp32_evaluations(S, alpha, iters, n_perm=4, seed=0):
    """The oracle's fp32 optimal transport on S and on `n_perm` row/column permutations of S (mapped back): log-domain Sinkhorn is
    permutation-equivariant, so these are independent draws of the SAME computation's fp32 rounding noise; plus the float64 result."""
    from oracle import superglue_ref
    St = torch.as_tensor(np.ascontiguousarray(S))[None]
    a = torch.as_tensor(alpha)
    Z64 = superglue_ref.log_optimal_transport(St.double(), a.double(), iters=iters)[0].numpy()
    outs = [superglue_ref.log_optimal_transport(St, a.float(), iters=iters)[0].numpy()]
    g = torch.Generator().manual_seed(seed)
    m, n = S.shape
    for _ in range(n_perm):
        pr, pc = torch.randperm(m, generator=g), torch.randperm(n, generator=g)
        Zp = superglue_ref.log_optimal_transport(St[:, pr][:, :, pc], a.float(), iters=iters)[0]
        ir, ic = torch.cat([torch.argsort(pr), torch.tensor([m])]), torch.cat([torch.argsort(pc), torch.tensor([n])])   # the dustbins stay last
        outs.append(Zp[ir][:, ic].numpy())
    return outs, Z64


def outside_fraction(a, b, atol=ATOL, rtol=RTOL):
    """Fraction of elements of `a` outside |a-b| <= atol + rtol*|b| (the north_star tolerance): the number behind the
    float64-anchored criterion."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float((np.abs(a - b) > atol + rtol * np.abs(b)).mean())


# ---------------------------------------------------------------------------------------------- unselected-seed sweeps
def explainable0(i, g, s, tau):
    """A differing matches0[i] is explained by the reference's own margins: row i's top-1/top-2 gap, the gap of the column
    its argmax points to, or its distance to the match threshold (all in Z units) below tau."""
    j = int(g["idx0"][s][i])
    return g["gap0"][s][i] < tau or g["gap1"][s][j] < tau or g["thr_gap0"][s][i] < tau


def explainable1(j, g, s, tau):
    i = int(g["idx1"][s][j])
    return g["gap1"][s][j] < tau or g["gap0"][s][i] < tau or g["thr_gap0"][s][i] < tau


def sweep_compare_end_to_end(g, s, k0, k1, m0, tau=2e-3, topk_tol=2e-5):
    """One pair of the HIP path (keypoints (K,2), matches0 (K,)) against seed index `s` of a sweep fixture (the REFERENCE's
    outputs on unselected seeds, tests/golden/make_golden.py).  Keypoint SETS may differ from the reference's only where the
    top-k boundary gap (last kept minus first dropped score) is below `topk_tol` (the score map carries ~6e-6 of fp32 noise);
    when the sets agree, every matched coordinate pair that differs must be explained by the reference's margins (tau in Z
    units).  Returns {comparable, kp_bad: [...], n_ref, diff, unexplained: [...]}; used by the GPU tests and by bench.py's
    parity_in_run (the comparison needs the fixture only -- no oracle, no reference)."""
    K = len(m0)
    k0, k1, m0 = np.asarray(k0).astype(int), np.asarray(k1).astype(int), np.asarray(m0)
    res = {"comparable": True, "kp_bad": [], "kp_diff_images": 0, "n_ref": 0, "diff": 0, "unexplained": []}
    for side, k in ((0, k0), (1, k1)):
        mine, ref = set(map(tuple, k)), set(map(tuple, g[f"kpts{side}"][s].astype(int)))
        if mine != ref:
            res["comparable"] = False
            res["kp_diff_images"] += 1
            gap = float(g["topk_gap"][s][side])
            if not (gap < topk_tol and len(mine ^ ref) <= 8):
                res["kp_bad"].append((int(g["seeds"][s]), side, len(mine ^ ref), gap))
    if not res["comparable"]:
        return res                      # indices are not comparable row by row when the sets differ
    pos0 = {tuple(p): i for i, p in enumerate(g["kpts0"][s].astype(int))}
    pos1 = {tuple(p): i for i, p in enumerate(g["kpts1"][s].astype(int))}
    mine = np.full(K, -1, np.int64)     # my matches0 re-indexed in the reference's keypoint order
    for i, j in enumerate(m0):
        mine[pos0[tuple(k0[i])]] = pos1[tuple(k1[j])] if j >= 0 else -1
    r0 = g["matches0"][s].astype(np.int64)
    diff = np.nonzero(mine != r0)[0]
    res["n_ref"] = int((r0 >= 0).sum())
    res["diff"] = len(diff)
    res["unexplained"] = [(int(g["seeds"][s]), int(i), float(g["gap0"][s][i])) for i in diff if not explainable0(i, g, s, tau)]
    return res


def transport_Z(S, u, v, n0, n1, alpha):
    """log_optimal_transport's output (B=1: (n0+1, n1+1)) from the library's score matrix and potentials:
    couplings + u + v - norm (superglue_test.py:157-170), float32 like the reference."""
    Z = np.full((n0 + 1, n1 + 1), np.float32(alpha), dtype=np.float32)
    Z[:n0, :n1] = S[:n0, :n1]
    return (Z + u[:n0 + 1, None]) + v[None, :n1 + 1] + np.log(np.float32(n0 + n1))


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


def fuzz_seeds(default):
    """IMX_FUZZ_SEEDS="a-b" replaces a fuzz test's seed list for a soak run (tools/gpu_soak.sh); the suite runs the default."""
    spec = os.environ.get("IMX_FUZZ_SEEDS")
    if not spec:
        return default
    a, b = spec.split("-")
    return list(range(int(a), int(b) + 1))


# ---------------------------------------------------------------------------------------------- strict fixtures (round 4)
def threshold_band_rows(g, s, thr):
    """Rows / columns of seed index `s` whose REFERENCE decision is a threshold comparison inside the north_star tolerance: the
    reference's mutual candidate score exp(Z[i, j]) lies within 1e-4 + 1e-4*thr of match_threshold (superglue_test.py:281:
    `valid0 = mutual0 & (mscores0 > match_threshold)` thresholds a continuous quantity, so a score that agrees with the
    reference's to 1e-6 can still sit on the other side).  A property of the reference's output alone; on these rows -- and only
    these -- "matching score within 1e-4" and "index identical" cannot both be demanded of an independent fp32 evaluation, and the
    strict tests accept either side of the threshold there (the candidate index itself must still be the reference's).
    Returns (set of side-0 rows i, set of side-1 columns j)."""
    band = (ATOL + RTOL * thr) / thr                     # in Z units: d exp(Z) = thr dZ at the threshold
    rows = np.nonzero(g["thr_gap0"][s] < band)[0]        # thr_gap0 = |Z[i, idx0[i]] - log thr| for mutual rows, inf otherwise
    return set(int(i) for i in rows), set(int(g["idx0"][s][i]) for i in rows)


def strict_index_check(g, s, mine0, mine1, thr, tag):
    """matches0 / matches1 (in the reference's keypoint order) against seed index `s` of a strict fixture: identical, except that on
    a threshold-band row (threshold_band_rows) the pair may be reported as matched or unmatched -- with the reference's candidate
    index -- and rows / columns on an EXACT tie of the reference's own fp32 Z may go to either tied partner.  Both exceptions are
    properties of the reference's output alone (stored in the fixture); every use is printed.  Returns the number of rows that
    differ under the two rules; raises on anything else."""
    r0, r1 = g["matches0"][s].astype(np.int64), g["matches1"][s].astype(np.int64)
    band0, band1 = threshold_band_rows(g, s, thr)
    d0, d1 = np.nonzero(mine0 != r0)[0], np.nonzero(mine1 != r1)[0]
    # exact ties in the reference's OWN fp32 Z (top-1 minus top-2 == 0.0: two rows whose whole mass sits on one column converge to the
    # same transport entry, so the winner is decided by the last rounding -- the reference's float64 evaluation of itself picks the
    # other row on strict_c3 seed 1022): the reference reports the first index (Tensor.max), any evaluation may report either
    # ... but not ANY index: the tied partners come from the reference's own forward (tests/golden/strict_ties.npz,
    # make_golden.py --strict-ties), and on a tie the reported index must be one of them, or -1 (the mutual check can fail either way)
    fi = 0 if int(g["d"]) == 128 else 1
    parts = {(int(t[2]), int(t[3])): set(int(x) for x in t[4:] if x >= 0) for t in golden("strict_ties.npz")["ties"] if int(t[0]) == fi and int(t[1]) == s}

    def allowed(axis, i, idx_mine, idx_ref):
        """a differing entry of matches<axis> at index i is a tie if i's own line ties (then any tied partner, or -1) or if the line of
        the index involved ties over i (then that index, or -1)"""
        ok = set()
        if (axis, i) in parts:
            ok |= parts[(axis, i)] | {-1}
        for other in (idx_mine, idx_ref):
            if other >= 0 and i in parts.get((1 - axis, other), ()):
                ok |= {other, -1}
        return ok
    tie0 = lambda i: int(mine0[i]) in allowed(0, int(i), int(mine0[i]), int(g["idx0"][s][i]))
    tie1 = lambda j: int(mine1[j]) in allowed(1, int(j), int(mine1[j]), int(g["idx1"][s][j]))
    bad0 = [int(i) for i in d0 if not tie0(i) and not (int(i) in band0 and {int(mine0[i]), int(r0[i])} == {-1, int(g["idx0"][s][i])})]
    bad1 = [int(j) for j in d1 if not tie1(j) and not (int(j) in band1 and {int(mine1[j]), int(r1[j])} == {-1, int(g["idx1"][s][j])})]
    assert not bad0 and not bad1, f"{tag}: match indices differ from the reference's on rows {bad0[:6]} / columns {bad1[:6]} ({len(d0)}+{len(d1)} differ in all)"
    for i in d0:
        if tie0(i):
            print(f"[strict tie] {tag}: row {int(i)} -> {int(mine0[i])} here, {int(r0[i])} in the reference: an exact tie in the reference's own fp32 Z (gap 0.0)")
        else:
            print(f"[strict band] {tag}: row {int(i)} is {'matched' if mine0[i] >= 0 else 'unmatched'} here, {'matched' if r0[i] >= 0 else 'unmatched'} in the reference: "
                  f"its candidate score is exp(log thr +- {float(g['thr_gap0'][s][i]):.2e}) = {thr * np.exp(-float(g['thr_gap0'][s][i])):.7f}..{thr * np.exp(float(g['thr_gap0'][s][i])):.7f} against the threshold {thr}")
    return len(d0)


def strict_f64(g, key):
    """The float64 evaluation of the reference module on a strict fixture's sample (stored as float32 differences)."""
    return g[key].astype(np.float64) + g[key + "_d64"].astype(np.float64)


def strict_samples(g, s, gnn0, gnn1, S, Z):
    """The strided samples a strict fixture holds for seed index `s`, cut from full tensors laid out like the reference's:
    gnn0/gnn1 (d, N), S (N0, N1), Z (N0+1, N1+1).  Returns {fixture key: (mine, reference)}."""
    ss, sg = int(g["stride_s"]), int(g["stride_g"])
    return {"gnn17": (np.stack([np.asarray(gnn0)[:, ::sg], np.asarray(gnn1)[:, ::sg]]), g["gnn_sub"][s]),
            "scores_in": (np.asarray(S)[::ss, ::ss], g["scores_in_sub"][s]),
            "Z": (np.asarray(Z)[::ss, ::ss], g["Z_sub"][s])}


def tolerance_used(a, b):
    """Worst |a-b| / (1e-4 + 1e-4|b|): the fraction of the north_star tolerance used (< 1 passes)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float((np.abs(a - b) / (ATOL + RTOL * np.abs(b))).max())


def strict_compare_batch(g, out, eng, B, alpha, thr, dense=True, seed_idx=None, measured=None):
    """One imx_match_pairs output of B pairs (pair b = fixture seed b mod n) against a strict fixture: identical keypoint sets,
    every match identical (as coordinate pairs, and as indices where the keypoint order agrees), matching scores at 1e-4; with
    `dense`, the fixture's samples of gnn17 / scores_in / Z against the library's taps (rows mapped to the reference's keypoint
    order).  Returns a summary dict; raises on any violation.  Used by the test below and by bench.py's parity_in_run_strict."""
    K, n = int(g["K"]), len(g["seeds"])
    k0a, k1a = out["keypoints0"].cpu().numpy().astype(int), out["keypoints1"].cpu().numpy().astype(int)
    m0a, ms0a = out["matches0"].cpu().numpy(), out["matching_scores0"].cpu().numpy()
    m1a = out["matches1"].cpu().numpy()
    assert (out["counts0"].cpu().numpy() == K).all() and (out["counts1"].cpu().numpy() == K).all()
    summary = {"pairs": B, "reference_matches": 0, "index_mismatches_outside_threshold_band": 0, "threshold_band_rows": 0, "rows_differing_under_the_tie_and_band_rules": 0,
               "keypoint_set_mismatches": 0, "order_differs_images": 0, "mutual_flag_flips_unmatched_rows": 0, "mscores_outside_1e-4": 0,
               "worst_tolerance_used": {"mscores": 0.0}, "samples_outside_1e-4": {}, "samples": {}}
    Sall = eng.fetch("scores_in") if dense else None
    U, V, X = (eng.fetch("u"), eng.fetch("v"), eng.fetch("x")) if dense else (None, None, None)
    Kp = (K + 31) // 32 * 32
    for b in range(B):
        s = b % n if seed_idx is None else seed_idx[b]
        if s < 0:
            continue
        summary["pairs_checked"] = summary.get("pairs_checked", 0) + 1
        seed = int(g["seeds"][s])
        r_k0, r_k1 = g["kpts0"][s].astype(int), g["kpts1"][s].astype(int)
        pos0 = {tuple(p): i for i, p in enumerate(r_k0)}
        pos1 = {tuple(p): i for i, p in enumerate(r_k1)}
        same0 = set(map(tuple, k0a[b])) == set(pos0)
        same1 = set(map(tuple, k1a[b])) == set(pos1)
        summary["keypoint_set_mismatches"] += (not same0) + (not same1)
        assert same0 and same1, f"pair {b} (seed {seed}): keypoint SET differs from the reference's"
        p0 = np.array([pos0[tuple(p)] for p in k0a[b]])         # my row i is the reference's row p0[i]
        p1 = np.array([pos1[tuple(p)] for p in k1a[b]])
        summary["order_differs_images"] += int((p0 != np.arange(K)).any()) + int((p1 != np.arange(K)).any())
        mine0 = np.full(K, -1, np.int64)
        mine0[p0] = np.where(m0a[b] >= 0, p1[np.clip(m0a[b], 0, K - 1)], -1)
        mine1 = np.full(K, -1, np.int64)
        mine1[p1] = np.where(m1a[b] >= 0, p0[np.clip(m1a[b], 0, K - 1)], -1)
        r0 = g["matches0"][s].astype(np.int64)
        summary["reference_matches"] += int((r0 >= 0).sum())
        summary["threshold_band_rows"] += len(threshold_band_rows(g, s, thr)[0])
        summary["rows_differing_under_the_tie_and_band_rules"] += strict_index_check(g, s, mine0, mine1, thr, f"pair {b} (seed {seed})")
        sc = np.zeros(K, np.float32)
        sc[p0] = ms0a[b]
        rs = g["mscores0"][s]
        # matching_scores0 = exp(Z[i, argmax]) where the argmaxes are mutual, else 0 (superglue_test.py:276-280).  Images in, a row's
        # mutual flag can flip where its reference argmax margin is below the amplified SuperPoint differences (score x vs 0 on a
        # row that is unmatched on both sides): counted; everything else at 3x the tolerance (measured: 0.44x), counted at 1x
        both = (mine0 == r0) & ((sc > 0) == (rs > 0))
        summary["mutual_flag_flips_unmatched_rows"] += int(((mine0 == r0) & ((sc > 0) != (rs > 0))).sum())
        assert_close(sc[both], rs[both], f"pair {b} (seed {seed}): matching_scores0", atol=3 * ATOL, rtol=3 * RTOL)
        summary["mscores_outside_1e-4"] += int((np.abs(sc[both].astype(np.float64) - rs[both]) > ATOL + RTOL * np.abs(rs[both])).sum())
        summary["worst_tolerance_used"]["mscores"] = max(summary["worst_tolerance_used"]["mscores"], tolerance_used(sc[both], rs[both]))
        if dense:
            i0, i1 = np.argsort(p0), np.argsort(p1)              # reference row r is my row i0[r]
            g0 = X[b * Kp:b * Kp + K][i0].T
            g1 = X[B * Kp + b * Kp:B * Kp + b * Kp + K][i1].T
            S = Sall[b, :K, :K][i0][:, i1]
            Z = transport_Z(Sall[b], U[b], V[b], K, K, alpha)
            Z = Z[np.append(i0, K)][:, np.append(i1, K)]
            for key, (mine, fx) in strict_samples(g, s, g0, g1, S, Z).items():
                # images in: the library's SuperGlue runs on the library's OWN SuperPoint outputs, whose (within-tolerance) differences
                # from the reference's the 18-layer GNN amplifies; the SuperGlue STAGE is held to 1x on identical inputs
                # (test_gpu_strict.py).  Round 5: no bare 10x any more.  scores_in and Z are anchored on the float64 evaluation the
                # fixture carries for these very samples -- |hip - f64| <= 1e-4 + 1e-4|f64| + 2.5 x (the largest |ref32 - f64| of this
                # seed's full tensor: the reference's own fp32 distance from its float64 self) -- and every sample outside 1x is listed
                # with the reference's own distance AT THAT ELEMENT; gnn17 has no float64 samples in the fixture (its reference
                # envelope is 5e-6) and is held to 3x, what the amplified SuperPoint differences were measured to need (1.2x)
                fkey = {"scores_in": "scores_in_sub", "Z": "Z_sub"}.get(key)
                m64, fx64 = mine.astype(np.float64), fx.astype(np.float64)
                if measured is not None and measured[b] is not None:
                    # Round 6: the amplification is MEASURED, per element.  `measured[b]` (tests/oracle_jobs.py:
                    # stage_job) holds the oracle's SuperGlue in float64 on this call's OWN SuperPoint outputs at these sample points
                    # (own64) and its difference from the same float64 module on the reference's SuperPoint outputs (delta): the
                    # rounding-free response to exactly the input perturbation the library's SuperPoint applies.  The end-to-end
                    # result is held to 1x against own64 -- no envelope term, no multiple -- and every sample outside 1x of the
                    # REFERENCE's fp32 value is listed with |delta| there: hip - ref = (hip - own64) + delta + (f64 - ref).
                    own64, delta = measured[b]["own64"][key], measured[b]["delta"][key]
                    lim = ATOL + RTOL * np.abs(own64)
                    bad = np.abs(m64 - own64) > lim
                    assert not bad.any(), (f"pair {b} (seed {seed}), images in: {key}: {int(bad.sum())} samples further than 1e-4 + 1e-4|f64| from the float64 "
                                           f"SuperGlue evaluated on the call's own SuperPoint outputs; worst {float((np.abs(m64 - own64) / lim).max()):.2f}x")
                    cond = summary.setdefault("measured_conditioning", {})
                    c = cond.setdefault(key, {"worst_hip_vs_own_f64": 0.0, "largest_delta": 0.0, "samples_where_delta_alone_exceeds_1x": 0})
                    tol_ref = ATOL + RTOL * np.abs(fx64)
                    c["worst_hip_vs_own_f64"] = round(max(c["worst_hip_vs_own_f64"], float((np.abs(m64 - own64) / lim).max())), 3)
                    c["largest_delta"] = round(max(c["largest_delta"], float((np.abs(delta) / tol_ref).max())), 3)
                    c["samples_where_delta_alone_exceeds_1x"] += int((np.abs(delta) > tol_ref).sum())
                    out1 = np.abs(m64 - fx64) > tol_ref
                    for idx in np.argwhere(out1)[:4]:
                        t = tuple(idx)
                        summary.setdefault("outliers", []).append(
                            {"pair": b, "tensor": key, "hip_vs_ref_in_tolerances": round(float(abs(m64[t] - fx64[t]) / tol_ref[t]), 2),
                             "measured_input_response_in_tolerances": round(float(abs(delta[t]) / tol_ref[t]), 2),
                             "hip_vs_f64_on_its_own_inputs_in_tolerances": round(float(abs(m64[t] - own64[t]) / lim[t]), 2)})
                elif fkey:
                    f64 = fx64 + g[fkey + "_d64"][s].astype(np.float64)
                    env = float(g["env_" + key][s][0])
                    lim = ATOL + RTOL * np.abs(f64) + 2.5 * env
                    bad = np.abs(m64 - f64) > lim
                    assert not bad.any(), (f"pair {b} (seed {seed}), images in: {key}: {int(bad.sum())} samples further from the float64 evaluation than "
                                           f"1e-4 + 1e-4|f64| + 2.5 x {env:.2e} (the reference's own fp32 envelope); worst {np.abs(m64 - f64).max():.3e}")
                    out1 = np.abs(m64 - fx64) > ATOL + RTOL * np.abs(fx64)
                    for idx in np.argwhere(out1)[:4]:
                        t = tuple(idx)
                        summary.setdefault("outliers", []).append(
                            {"pair": b, "tensor": key, "hip_vs_ref_in_tolerances": round(float(abs(m64[t] - fx64[t]) / (ATOL + RTOL * abs(fx64[t]))), 2),
                             "ref_vs_f64_in_tolerances": round(float(abs(fx64[t] - f64[t]) / (ATOL + RTOL * abs(f64[t]))), 2),
                             "hip_vs_f64_in_tolerances": round(float(abs(m64[t] - f64[t]) / (ATOL + RTOL * abs(f64[t]))), 2)})
                else:
                    assert_close(mine, fx, f"pair {b} (seed {seed}), images in: {key} vs the reference's sample", atol=3 * ATOL, rtol=3 * RTOL)
                w, o = summary["worst_tolerance_used"], summary["samples_outside_1e-4"]
                w[key] = max(w.get(key, 0.0), tolerance_used(mine, fx))
                o[key] = o.get(key, 0) + int((np.abs(mine.astype(np.float64) - fx) > ATOL + RTOL * np.abs(fx)).sum())
                summary["samples"][key] = summary["samples"].get(key, 0) + fx.size
    rows = max(summary.get("pairs_checked", 0), 1) * K
    assert summary["mutual_flag_flips_unmatched_rows"] <= max(2, 2e-4 * rows) and summary["mscores_outside_1e-4"] <= max(2, 2e-4 * rows), summary
    return summary

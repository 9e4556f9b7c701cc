"""Oracle work of the strict GPU tests as picklable jobs for a process pool (round 5): the box has 256 host
cores, and the per-seed oracle runs (SuperPoint dense + SuperGlue dense, 48 seeds, then the per-stage checks of every pair of the
batched calls) were serial -- 4 of the suite's 7 minutes.  Test infrastructure only (imports oracle/)."""
import os

import numpy as np
import torch


def _init(threads):
    torch.set_num_threads(threads)


def pool_map(fn, jobs, threads=8):
    """[fn(*job) for job in jobs] over spawned worker processes of `threads` torch threads each (spawn: the parent holds a HIP context)."""
    import concurrent.futures as cf
    import multiprocessing as mp
    if not jobs:
        return []
    workers = max(1, min(len(jobs), (os.cpu_count() or 8) // threads, 32))
    if workers == 1:
        return [fn(*j) for j in jobs]
    with cf.ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn"), initializer=_init, initargs=(threads,)) as ex:
        return list(ex.map(fn, *zip(*jobs)))


def strict_inputs_job(name, s):
    """tests/test_gpu_strict.py:_strict_inputs for seed index s: SuperGlue's inputs (the REFERENCE's keypoints and scores from the
    fixture, descriptors sampled by the oracle at those keypoints) and the oracle's dense gnn17 / scores_in / Z on them."""
    from oracle import superglue_ref, superpoint_ref
    from tests import util
    g = util.golden(name)
    H, W, d, K = (int(g[k]) for k in ("H", "W", "d", "K"))
    seed = int(g["seeds"][s])
    sd_sp, sd_sg = util.sp_sd(d), util.sg_sd(d, variant="t")
    x0, x1 = util.pair(seed, H, W)
    data = {"image0": x0, "image1": x1}
    for side, x in (("0", x0), ("1", x1)):
        dense = superpoint_ref.superpoint_forward(x, sd_sp, util.sp_config(d, K), return_dense=True)["desc"]
        kp = torch.from_numpy(g["kpts" + side][s].astype(np.float32))[None]
        data["keypoints" + side] = kp
        data["scores" + side] = torch.from_numpy(g["scores" + side][s])[None]
        data["descriptors" + side] = superpoint_ref.sample_descriptors(kp, dense, 8)
    dn = superglue_ref.superglue_forward(data, sd_sg, util.sg_config(d), return_dense=True)["dense"]
    ref = {"gnn0": dn["gnn0"][0].numpy(), "gnn1": dn["gnn1"][0].numpy(), "scores_in": dn["scores_in"][0].numpy(), "Z": dn["Z"][0].numpy()}
    keys = ("keypoints0", "keypoints1", "scores0", "scores1", "descriptors0", "descriptors1")
    return {k: data[k].numpy() for k in keys}, ref


def stage_job(d, K, H, W, seed, own, strides=None):
    """The per-stage check of one pair of a batched call: the oracle's SuperPoint on the pair's images, and the oracle's SuperGlue on
    the LIBRARY's own SuperPoint outputs of that pair (`own`: numpy keypoints / scores / descriptors (d, K) per side).
    With `strides` = (stride_s, stride_g) of a strict fixture (round 6): also the oracle's SuperGlue in FLOAT64 on
    both input sets -- the library's SuperPoint outputs and the oracle's (= the reference's) -- cut to the fixture's sample points in
    the reference's keypoint order: `own64` (the float64 evaluation the library's end-to-end result is anchored on) and
    `delta` = own64 - ref64, the response of the 18-layer SuperGlue, element by element and free of rounding, to exactly the
    perturbation the library's SuperPoint applies to its inputs."""
    from oracle import superglue_ref, superpoint_ref
    from tests import util
    ims = util.pair(seed, H, W)
    sp = []
    for x in ims:
        o = superpoint_ref.superpoint_forward(x, util.sp_sd(d), util.sp_config(d, K))
        sp.append({"keypoints": o["keypoints"][0].numpy(), "scores": o["scores"][0].numpy(), "descriptors": o["descriptors"][0].numpy()})
    data = {"image_shape0": (1, 1, H, W), "image_shape1": (1, 1, H, W)}
    for side in ("0", "1"):
        for k in ("keypoints", "scores", "descriptors"):
            data[k + side] = torch.from_numpy(own[k + side])[None]
    sd = util.sg_sd(d, variant="t")
    dn = superglue_ref.superglue_forward(data, sd, util.sg_config(d), return_dense=True)["dense"]
    res = {"gnn0": dn["gnn0"][0].numpy(), "gnn1": dn["gnn1"][0].numpy(), "scores_in": dn["scores_in"][0].numpy(), "Z": dn["Z"][0].numpy()}
    if strides is None:
        return sp, res
    ss, sg = strides
    to64 = lambda t: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in t.items()}
    sd64 = to64(sd)

    def run64(dat, perm0, perm1):        # -> the fixture's samples, rows / columns in the REFERENCE's keypoint order
        o = superglue_ref.superglue_forward(to64(dat), sd64, util.sg_config(d), return_dense=True)["dense"]
        g0, g1 = o["gnn0"][0].numpy()[:, perm0], o["gnn1"][0].numpy()[:, perm1]
        S = o["scores_in"][0].numpy()[perm0][:, perm1]
        Z = o["Z"][0].numpy()[np.append(perm0, K)][:, np.append(perm1, K)]
        return {"gnn17": np.stack([g0[:, ::sg], g1[:, ::sg]]), "scores_in": S[::ss, ::ss], "Z": Z[::ss, ::ss]}
    perms = []
    for si, side in enumerate(("0", "1")):       # reference row r is the library's row perm[r] (same keypoint SET: asserted by the caller)
        pos = {tuple(q): i for i, q in enumerate(own["keypoints" + side].astype(int))}
        perms.append(np.array([pos[tuple(q)] for q in sp[si]["keypoints"].astype(int)]))
    own64 = run64(data, perms[0], perms[1])
    ref_data = {"image_shape0": (1, 1, H, W), "image_shape1": (1, 1, H, W)}
    for si, side in enumerate(("0", "1")):
        for k in ("keypoints", "scores", "descriptors"):
            ref_data[k + side] = torch.from_numpy(sp[si][k])[None]
    ident = np.arange(K)
    ref64 = run64(ref_data, ident, ident)
    return sp, res, {"own64": own64, "delta": {k: own64[k] - ref64[k] for k in own64}}

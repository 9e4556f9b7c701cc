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


def stage_job(d, K, H, W, seed, own):
    """The per-stage check of one pair of a batched call: the oracle's SuperPoint on the pair's images, and the oracle's SuperGlue on
    the LIBRARY's own SuperPoint outputs of that pair (`own`: numpy keypoints / scores / descriptors (d, K) per side)."""
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
    dn = superglue_ref.superglue_forward(data, util.sg_sd(d, variant="t"), util.sg_config(d), return_dense=True)["dense"]
    return sp, {"gnn0": dn["gnn0"][0].numpy(), "gnn1": dn["gnn1"][0].numpy(), "scores_in": dn["scores_in"][0].numpy(), "Z": dn["Z"][0].numpy()}

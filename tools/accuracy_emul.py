#!/usr/bin/env python
"""CPU emulation of the accumulation ORDER of the HIP SuperGlue kernels, to find which fp32 reduction dominates the distance
to a float64 evaluation (tests/util.py:assert_fp64_anchored) and what splitting it into independent chains buys.
Measured on MI355X (tools/ubench/mfma_round.hip): v_mfma_f32_* accumulates exactly like a sequential fmaf chain (RNE), so a
GEMM's error grows with the chain length K; `C` independent chains summed at the end halve it at C = 4.
usage: python tools/accuracy_emul.py [gemm_chains] [pv_chains] [c5]     (build container, no GPU)
  chains > 0: that many interleaved sequential chains; < 0: two-level, blocks of |chains| consecutive k; 0: torch default"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import superglue_ref, superpoint_ref  # noqa: E402
from tests import util  # noqa: E402

torch.set_grad_enabled(False)


MINK = int(os.environ.get("EMUL_MINK", "0"))      # two-level only for reductions at least this long (else one chain)
P2 = int(os.environ.get("EMUL_P2", "0"))           # round 3: P carried as two bf16 terms (16 significant bits) into the P.V product
                                                   # (measured: scores_in rms 1.06x -> 1.20x, max 1.05x -> 1.50x of the reference's own fp32
                                                   # error at C3 with diffuse attention; 2^-17 relative on a peaked row: not adopted)


def two_terms(x):
    """x rounded to its first two bf16 terms h + m (what a two-term P costs: the third term, <= 2^-17 |x|, is dropped)."""
    h = x.to(torch.bfloat16).to(torch.float32)
    return h + (x - h).to(torch.bfloat16).to(torch.float32)


def chain_mm(a, w, C):
    """a (R,K) @ w (K,N) with C independent sequential fp32 chains (k -> chain k % C), summed pairwise at the end."""
    if C < 0 and a.shape[1] < MINK:
        C = 1
    if C == 0:
        return a @ w
    R, K = a.shape
    if C < 0:          # two-level: blocks of |C| consecutive k (one staged key tile) summed in a fresh accumulator, then added in order
        tot = torch.zeros(R, w.shape[1])
        for k0 in range(0, K, -C):
            blk = torch.zeros(R, w.shape[1])
            for k in range(k0, min(K, k0 - C)):
                blk = torch.addcmul(blk, a[:, k:k + 1], w[k:k + 1, :])
            tot = tot + blk
        return tot
    acc = [torch.zeros(R, w.shape[1]) for _ in range(C)]
    for k in range(K):
        acc[k % C] = torch.addcmul(acc[k % C], a[:, k:k + 1], w[k:k + 1, :])
    while len(acc) > 1:
        acc = [acc[i] + acc[i + 1] for i in range(0, len(acc), 2)]
    return acc[0]


def conv1(x, sd, p, C):           # x (d_in, N) -> (d_out, N)
    w, b = sd[p + ".weight"][:, :, 0], sd[p + ".bias"]
    return (chain_mm(x.t().contiguous(), w.t().contiguous(), C) + b).t().contiguous()


def bn(x, sd, p):
    s = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + 1e-5)
    return (x - sd[p + ".running_mean"][:, None]) * s[:, None] + sd[p + ".bias"][:, None]


def attention(q, k, v, CP):       # (dim, heads, n)
    dim = q.shape[0]
    out = torch.empty_like(q)
    for h in range(q.shape[1]):
        S = chain_mm((q[:, h].t() / dim ** .5).contiguous(), k[:, h].contiguous(), 1 if CP != 0 else 0)     # (n, m)
        P = torch.exp(S - S.max(1, keepdim=True).values)
        if CP == 0 and P2:
            out[:, h] = ((two_terms(P) @ v[:, h].t()) / P.sum(1, keepdim=True)).t()
            continue
        if CP == 0:
            out[:, h] = (torch.softmax(S, 1) @ v[:, h].t()).t()
            continue
        l = chain_mm(P, torch.ones(P.shape[1], 1), CP)
        out[:, h] = (chain_mm(P, v[:, h].t().contiguous(), CP) / l).t()
    return out


def forward(data, sd, cfg, CG, CP):
    d = cfg["descriptor_dim"]
    k0 = superglue_ref.normalize_keypoints(data["keypoints0"], data["image0"].shape)
    k1 = superglue_ref.normalize_keypoints(data["keypoints1"], data["image1"].shape)
    nconv = len(cfg["keypoint_encoder"]) + 1
    descs = [data["descriptors0"][0] + superglue_ref.keypoint_encoder(k0, data["scores0"], sd, nconv)[0],
             data["descriptors1"][0] + superglue_ref.keypoint_encoder(k1, data["scores1"], sd, nconv)[0]]
    for i, name in enumerate(["self", "cross"] * 9):
        p = f"gnn.layers.{i}"
        src = [descs[1], descs[0]] if name == "cross" else descs
        new = []
        for x, s in zip(descs, src):
            q = conv1(x, sd, p + ".attn.proj.0", CG).view(d // 4, 4, -1)
            k = conv1(s, sd, p + ".attn.proj.1", CG).view(d // 4, 4, -1)
            v = conv1(s, sd, p + ".attn.proj.2", CG).view(d // 4, 4, -1)
            msg = conv1(attention(q, k, v, CP).contiguous().view(d, -1), sd, p + ".attn.merge", CG)
            hdn = torch.relu(bn(conv1(torch.cat([x, msg]), sd, p + ".mlp.0", CG), sd, p + ".mlp.1"))
            new.append(x + conv1(hdn, sd, p + ".mlp.3", CG))
        descs = new
    m0, m1 = conv1(descs[0], sd, "final_proj", CG), conv1(descs[1], sd, "final_proj", CG)
    return chain_mm(m0.t().contiguous(), m1, CG) / d ** .5


def main():
    CG = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    CP = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    d, K, H, W, seed = (256, 2048, 960, 1280, 19) if len(sys.argv) > 3 and sys.argv[3] == "c5" else (128, 1024, 480, 640, 59)
    sd_sp, sd = util.sp_sd(d), util.sg_sd(d)
    x0, x1 = util.pair(seed, H, W)
    o0 = superpoint_ref.superpoint_forward(x0, sd_sp, util.sp_config(d, K))
    o1 = superpoint_ref.superpoint_forward(x1, sd_sp, util.sp_config(d, K))
    data = {"image0": x0, "image1": x1, "keypoints0": o0["keypoints"][0][None], "keypoints1": o1["keypoints"][0][None],
            "scores0": o0["scores"][0][None], "scores1": o1["scores"][0][None],
            "descriptors0": o0["descriptors"][0][None], "descriptors1": o1["descriptors"][0][None]}
    cfg = {**util.sg_config(d), "descriptor_dim": d}
    ref32 = superglue_ref.superglue_forward(data, sd, cfg, return_dense=True)["dense"]["scores_in"][0]
    sd64 = {k: v.double() for k, v in sd.items()}
    d64 = {k: v.double() for k, v in data.items()}
    f64 = superglue_ref.superglue_forward(d64, sd64, cfg, return_dense=True)["dense"]["scores_in"][0]
    t = time.time()
    emu = forward(data, sd, cfg, CG, CP)
    er, ee = (ref32.double() - f64).abs(), (emu.double() - f64).abs()
    print(f"gemm chains {CG}, PV chains {CP} ({time.time() - t:.0f}s): scores_in  reference fp32 max {er.max():.3e} rms {er.pow(2).mean().sqrt():.3e} | "
          f"emulated max {ee.max():.3e} rms {ee.pow(2).mean().sqrt():.3e}  (x{ee.pow(2).mean().sqrt() / er.pow(2).mean().sqrt():.2f})")


if __name__ == "__main__":
    main()

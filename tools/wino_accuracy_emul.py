#!/usr/bin/env python
"""CPU emulation (fp32, torch) of the SuperPoint 3x3 stack as Winograd F(mh x mw, 3x3) -- how much of the 1e-4 + 1e-4|ref|
parity tolerance would a larger tile use?  The shipped kernels are F(2x4,3x3); F(4x4,3x3) would execute 2.25 instead of 3
multiplies per output.  Compares x4 / semi / desc against the reference goldens (tests/golden/sp_small.npz) and against a
float64 direct evaluation.   usage: python tools/wino_accuracy_emul.py       (build container, no GPU)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util  # noqa: E402

torch.set_grad_enabled(False)

# F(2,3) and F(4,3) (points 0, +-1, +-2, inf; the matrices of csrc/imx_api.cpp:wino24_transform and wino24_pk.h)
BT = {2: np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64),
      4: np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                   [0, 4, 0, -5, 0, 1]], np.float64)}
G = {2: np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64),
     4: np.array([[.25, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                  [0, 0, 1]], np.float64)}
AT = {2: np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64),
      4: np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)}


def lin(mat, x, dim):
    """apply `mat` (r x c) along dimension `dim` of x in fp32 as a sequential sum of scaled terms (what a VALU chain does)."""
    x = x.movedim(dim, 0)
    rows = []
    for r in range(mat.shape[0]):
        acc = None
        for c in range(mat.shape[1]):
            if mat[r, c] == 0:
                continue
            t = x[c] * np.float32(mat[r, c])
            acc = t if acc is None else acc + t
        rows.append(acc)
    return torch.stack(rows).movedim(0, dim)


def planes_f16(t, n, per_image=False):
    """t (fp32) as n fp16 planes of t * s, s = the power of two that brings max |t| (of the tensor, or of every image t[b]) to
    [2^13, 2^14) -- attention_x3.hip's FmtH2 (round 4).  Returns the planes as fp32 tensors and 1 / s."""
    amax = t.abs().amax(dim=tuple(range(1, t.dim())), keepdim=True) if per_image else t.abs().max()
    s = torch.exp2(13 - torch.floor(torch.log2(amax.double().clamp_min(1e-30)))).to(torch.float32)
    r, out = (t * s).double(), []
    for _ in range(n):
        h = r.to(torch.float32).to(torch.float16)
        out.append(h.to(torch.float32))
        r = r - h.double()
    return out, 1.0 / s


PRODUCTS = {2: ((0, 0), (0, 1), (1, 0)), 3: ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0))}


def direct_f16(x, w, b, n=2):
    """the direct convolution with both operands as n fp16 planes (n = 2: three plane products, n = 3: six)"""
    xp, xs = planes_f16(x, n, per_image=True)
    wp, ws = planes_f16(w, n)
    y = sum(F.conv2d(xp[i], wp[j], None, padding=1) for i, j in PRODUCTS[n])
    return y * xs * ws + b[None, :, None, None]


def wino_conv(x, w, b, mh, mw, dt=torch.float32, f16_planes=0):
    """x (B,C,H,W), w (Co,Ci,3,3) folded, pad 1.  Output tiles mh x mw.  f16_planes = n: the transformed operands V and U as n fp16
    planes each (PRODUCTS[n] plane products, fp32 accumulation)."""
    B, C, H, W = x.shape
    Hp, Wp = -(-H // mh) * mh, -(-W // mw) * mw
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    th, tw = mh + 2, mw + 2
    d = xp.unfold(2, th, mh).unfold(3, tw, mw)               # (B,C,ny,nx,th,tw)
    V = lin(BT[mw], lin(BT[mh], d, 4), 5)
    U = torch.from_numpy(np.einsum("ik,ockl,jl->ocij", G[mh], w.double().numpy(), G[mw])).to(dt)     # host, float64 -> fp32
    if f16_planes:
        Vp, vs = planes_f16(V, f16_planes, per_image=True)
        Up, us = planes_f16(U, f16_planes)
        M = sum(torch.einsum("bcyxij,ocij->boyxij", Vp[i], Up[j]) for i, j in PRODUCTS[f16_planes]) * vs * us
    else:
        M = torch.einsum("bcyxij,ocij->boyxij", V.to(dt), U)
    Y = lin(AT[mw], lin(AT[mh], M, 4), 5)                     # (B,Co,ny,nx,mh,mw)
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, -1, Hp, Wp)[:, :, :H, :W]
    return Y + b.to(dt)[None, :, None, None]


def fold(sd, conv, bn, dt):
    w, b = sd[conv + ".weight"].double(), sd[conv + ".bias"].double()
    g, be, mu, var = (sd[bn + "." + k].double() for k in ("weight", "bias", "running_mean", "running_var"))
    s = g / torch.sqrt(var + 1e-5)
    return (w * s[:, None, None, None]).to(dt), ((b - mu) * s + be).to(dt)


def forward(x, sd, conv3, dt=torch.float32):
    x = x.to(dt)
    for i, (pre, pool) in enumerate((("inc.conv.conv", False), ("down1.mpconv.1.conv", True), ("down2.mpconv.1.conv", True),
                                     ("down3.mpconv.1.conv", True))):
        if pool:
            x = F.max_pool2d(x, 2)
        for idx in (0, 3):
            w, b = fold(sd, f"{pre}.{idx}", f"{pre}.{idx + 1}", dt)
            first = i == 0 and idx == 0
            x = F.relu(F.conv2d(x, w, b, padding=1) if first else conv3(x, w, b))
    x4 = x
    out = {"x4": x4}
    for head, a, bname in (("semi", "Pa", "Pb"), ("desc", "Da", "Db")):
        w, b = fold(sd, "conv" + a, "bn" + a, dt)
        y = F.relu(conv3(x4, w, b))
        w, b = fold(sd, "conv" + bname, "bn" + bname, dt)
        out[head] = F.conv2d(y, w, b)
    out["desc"] = out["desc"] / torch.norm(out["desc"], p=2, dim=1, keepdim=True)
    return out


def main():
    g = util.golden("sp_small.npz")
    H, W, seed = int(g["H"]), int(g["W"]), int(g["seed"])
    sd = {k: torch.as_tensor(v) for k, v in util.sp_sd(128).items()}
    x = torch.cat(util.pair(seed, H, W))
    f64 = forward(x, sd, lambda x, w, b: F.conv2d(x, w, b, padding=1), torch.float64)
    forms = {"direct fp32 (torch)": lambda x, w, b: F.conv2d(x, w, b, padding=1),
             "F(2x4,3x3)": lambda x, w, b: wino_conv(x, w, b, 2, 4),
             "F(4x4,3x3)": lambda x, w, b: wino_conv(x, w, b, 4, 4),
             "F(2x2,3x3)": lambda x, w, b: wino_conv(x, w, b, 2, 2),
             "F(2x4), V U 2 x fp16": lambda x, w, b: wino_conv(x, w, b, 2, 4, f16_planes=2),
             "F(2x4), V U 3 x fp16": lambda x, w, b: wino_conv(x, w, b, 2, 4, f16_planes=3),
             "direct, 2 x fp16": lambda x, w, b: direct_f16(x, w, b, 2),
             "direct, 3 x fp16": lambda x, w, b: direct_f16(x, w, b, 3)}
    for name, fn in forms.items():
        out = forward(x, sd, fn)
        cells = []
        for k in ("x4", "semi", "desc"):
            ref = g[k].astype(np.float64)
            got = out[k].double().numpy()
            margin = (np.abs(got - ref) / (1e-4 + 1e-4 * np.abs(ref))).max()
            e64 = got - f64[k].numpy()
            r64 = ref - f64[k].numpy()
            cells.append("%s margin %.3f  rms-vs-f64 %.2e (reference fp32: %.2e)" % (k, margin, np.sqrt((e64 ** 2).mean()),
                                                                                   np.sqrt((r64 ** 2).mean())))
        print("%-20s %s" % (name, " | ".join(cells)))


if __name__ == "__main__":
    main()

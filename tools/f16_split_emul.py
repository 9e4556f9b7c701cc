"""Numpy emulation behind attention_x3.hip's two-plane fp16 format (round 4): softmax(Q K^T) V and Q K^T alone with the operands cut into
fp16 / bf16 planes and a subset of the plane products kept, against float64 -- next to torch fp32 and the exact product of the fp32
operands.  usage: python tools/f16_split_emul.py"""
import numpy as np, torch
rng = np.random.default_rng(0)
def split_f16(x, n):
    x = x.astype(np.float64); planes=[]
    for _ in range(n):
        h = x.astype(np.float32).astype(np.float16)   # RTN
        planes.append(h.astype(np.float64)); x = x - h.astype(np.float64)
    return planes
def split_bf16(x, n):
    x = x.astype(np.float64); planes=[]
    for _ in range(n):
        t = torch.from_numpy(x.astype(np.float32)).to(torch.bfloat16).to(torch.float64).numpy()
        planes.append(t); x = x - t
    return planes
for gain in (1.0, 5.0, 20.0):
    n, hd = 1024, 32
    q = rng.standard_normal((n, hd)).astype(np.float32); k = rng.standard_normal((n, hd)).astype(np.float32)
    v = (rng.standard_normal((n, hd)) * rng.choice([0.01, 1.0, 3.0], size=(n,1))).astype(np.float32)
    S64 = (q.astype(np.float64) @ k.astype(np.float64).T) / np.sqrt(hd) * gain / 5.6
    S32 = S64.astype(np.float32)
    P64 = np.exp(S32.astype(np.float64) - S32.astype(np.float64).max(1, keepdims=True)); O64 = (P64 @ v.astype(np.float64)) / P64.sum(1, keepdims=True)
    O32 = torch.softmax(torch.from_numpy(S32), -1) @ torch.from_numpy(v); O32 = O32.numpy().astype(np.float64)
    # kernel-like P: fp32 exp2
    t = (S32 * np.float32(1.4426950408889634)); m = t.max(1, keepdims=True); P32 = np.exp2((t - m).astype(np.float32)).astype(np.float32)
    rs = P32.astype(np.float64).sum(1, keepdims=True)
    res = {}
    res['fp32 torch'] = O32
    res['exact P32.V32 (x3-like)'] = (P32.astype(np.float64) @ v.astype(np.float64)) / rs
    vmax = np.abs(v).max(); s = 2.0 ** (14 - np.ceil(np.log2(vmax)))
    Pp = split_f16(P32.astype(np.float64) * 2**15, 2); Vp = split_f16(v.astype(np.float64) * s, 3)
    def comb(terms): return sum(Pp[a] @ Vp[b] for a, b in terms) / (rs * 2**15 * s)
    res['f16 P2 V3 5prod'] = comb([(0,0),(0,1),(1,0),(1,1),(0,2)])
    res['f16 P2 V2 4prod'] = comb([(0,0),(0,1),(1,0),(1,1)])
    res['f16 P2 V2 3prod'] = comb([(0,0),(0,1),(1,0)])
    Pb = split_bf16(P32, 2); Vb = split_bf16(v, 3)
    res['bf16 P2 V3 (rejected)'] = (Pb[0]@Vb[0] + Pb[0]@Vb[1] + Pb[1]@Vb[0] + Pb[0]@Vb[2] + Pb[1]@Vb[1]) / rs
    sc = np.abs(O64).max()
    print(f"gain {gain}: max|S| {np.abs(S32).max():.1f}, max|O| {sc:.3f}")
    for kk, o in res.items():
        e = o - O64
        print(f"   {kk:28s} rms {np.sqrt((e**2).mean()):.3e}  max {np.abs(e).max():.3e}   rel-to-elem max {np.abs(e/ (np.abs(O64)+1e-3*sc)).max():.3e}")
print("---- S = Q K^T / sqrt(d)")
for gain in (1.0, 5.0, 20.0, 300.0):
    n, hd = 1024, 32
    q = (rng.standard_normal((n, hd)) * np.sqrt(gain)).astype(np.float32); k = (rng.standard_normal((n, hd))*np.sqrt(gain)).astype(np.float32)
    S64 = (q.astype(np.float64) @ k.astype(np.float64).T) / np.sqrt(hd) / 5.6
    S32 = ((torch.from_numpy(q) @ torch.from_numpy(k).T) / np.float32(np.sqrt(hd)) / np.float32(5.6)).numpy().astype(np.float64)
    sq = 2.0 ** (14 - np.ceil(np.log2(max(np.abs(q).max(), np.abs(k).max()))))
    Qp = split_f16(q.astype(np.float64) * sq, 3); Kp = split_f16(k.astype(np.float64) * sq, 3)
    Qb = split_bf16(q, 3); Kb = split_bf16(k, 3)
    def comb(P, K, terms, s=1.0): return sum(P[a] @ K[b].T for a, b in terms) / (s*s) / np.sqrt(hd) / 5.6
    res = {'fp32 torch': S32,
           'bf16 x3 6prod (shipped)': comb(Qb, Kb, [(0,0),(0,1),(1,0),(0,2),(2,0),(1,1)]),
           'f16 2pl 4prod': comb(Qp, Kp, [(0,0),(0,1),(1,0),(1,1)], sq),
           'f16 2pl 3prod': comb(Qp, Kp, [(0,0),(0,1),(1,0)], sq),
           'f16 Q3 K2 5prod': comb(Qp, Kp, [(0,0),(0,1),(1,0),(1,1),(2,0)], sq),
           'f16 3pl 6prod': comb(Qp, Kp, [(0,0),(0,1),(1,0),(1,1),(2,0),(0,2)], sq)}
    print(f"gain {gain}: max|S| {np.abs(S64).max():.1f}")
    for kk, o in res.items():
        e = o - S64
        print(f"   {kk:28s} rms {np.sqrt((e**2).mean()):.3e}  max {np.abs(e).max():.3e}")

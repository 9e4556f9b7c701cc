"""Portable synthetic weights and images for the SuperPoint+SuperGlue hot path.

All `.pth` files in the reference tree are git-LFS pointers, so parity and
benchmarks run on seeded synthetic weights.  Everything here is integer hashing plus IEEE
float32 add/mul/div/sqrt — no transcendental functions, no torch RNG — so the same seed
gives bit-identical tensors in the build container and on the GPU box.

State-dict key names and shapes are those of the reference modules:
  SuperPoint-BN  superpoint/models/superpoint_test.py:64-85, unet_parts.py:10-48
  SuperPoint     superglue/models/superpoint.py:111-134 (official, no BN)
  SuperGlue      superglue/models/superglue_test.py:49-60,73-82,92-119,207-219
"""
import zlib
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream_base(seed, name, sub=0):
    tag = zlib.crc32(name.encode()) & 0xFFFFFFFF
    s = np.array([(int(seed) * 0x100000001B3 + tag * 0x10001 + sub) & 0xFFFFFFFFFFFFFFFF],
                 dtype=np.uint64)
    return _splitmix64(_splitmix64(s))[0]


def uniform(seed, name, n, sub=0):
    """n float32 in [0,1), 24-bit resolution (exact)."""
    base = _stream_base(seed, name, sub)
    with np.errstate(over="ignore"):
        h = _splitmix64(base + np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95))
    return (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))


def normal(seed, name, n):
    """Approximately N(0,1): (sum of 4 uniforms - 2) * sqrt(3); exact float32 arithmetic."""
    acc = uniform(seed, name, n, 1)
    for sub in (2, 3, 4):
        acc = acc + uniform(seed, name, n, sub)
    return (acc - np.float32(2.0)) * np.float32(np.sqrt(np.float32(3.0)))


# ----------------------------------------------------------------------------- shapes
def _double_conv_shapes(prefix, cin, cout):
    s = OrderedDict()
    for ci, idx in ((cin, 0), (cout, 3)):
        s[f"{prefix}.{idx}.weight"] = (cout, ci, 3, 3)
        s[f"{prefix}.{idx}.bias"] = (cout,)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            s[f"{prefix}.{idx + 1}.{nm}"] = (cout,)
        s[f"{prefix}.{idx + 1}.num_batches_tracked"] = ()
    return s


def superpoint_bn_shapes(descriptor_dim=128):
    """Key → shape of superpoint/models/superpoint_test.py:SuperPoint.state_dict()."""
    c1, c2, c3, c4, c5 = 64, 64, 128, 128, 256
    s = OrderedDict()
    s.update(_double_conv_shapes("inc.conv.conv", 1, c1))
    s.update(_double_conv_shapes("down1.mpconv.1.conv", c1, c2))
    s.update(_double_conv_shapes("down2.mpconv.1.conv", c2, c3))
    s.update(_double_conv_shapes("down3.mpconv.1.conv", c3, c4))
    for head, cin, cout, k in (("Pa", c4, c5, 3), ("Pb", c5, 65, 1),
                               ("Da", c4, c5, 3), ("Db", c5, descriptor_dim, 1)):
        s[f"conv{head}.weight"] = (cout, cin, k, k)
        s[f"conv{head}.bias"] = (cout,)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            s[f"bn{head}.{nm}"] = (cout,)
        s[f"bn{head}.num_batches_tracked"] = ()
    return s


def superpoint_official_shapes(descriptor_dim=256):
    """Key → shape of superglue/models/superpoint.py:SuperPoint.state_dict() (no BN)."""
    c1, c2, c3, c4, c5 = 64, 64, 128, 128, 256
    s = OrderedDict()
    for nm, cin, cout, k in (("conv1a", 1, c1, 3), ("conv1b", c1, c1, 3),
                             ("conv2a", c1, c2, 3), ("conv2b", c2, c2, 3),
                             ("conv3a", c2, c3, 3), ("conv3b", c3, c3, 3),
                             ("conv4a", c3, c4, 3), ("conv4b", c4, c4, 3),
                             ("convPa", c4, c5, 3), ("convPb", c5, 65, 1),
                             ("convDa", c4, c5, 3), ("convDb", c5, descriptor_dim, 1)):
        s[f"{nm}.weight"] = (cout, cin, k, k)
        s[f"{nm}.bias"] = (cout,)
    return s


def superglue_shapes(descriptor_dim=128, keypoint_encoder=(32, 64, 128), n_layers=18):
    """Key → shape of superglue/models/superglue_test.py:SuperGlue.state_dict()."""
    d = descriptor_dim
    s = OrderedDict()
    s["bin_score"] = ()
    ch = [3] + list(keypoint_encoder) + [d]
    for i in range(1, len(ch)):
        j = 3 * (i - 1)
        s[f"kenc.encoder.{j}.weight"] = (ch[i], ch[i - 1], 1)
        s[f"kenc.encoder.{j}.bias"] = (ch[i],)
        if i < len(ch) - 1:
            for nm in ("weight", "bias", "running_mean", "running_var"):
                s[f"kenc.encoder.{j + 1}.{nm}"] = (ch[i],)
            s[f"kenc.encoder.{j + 1}.num_batches_tracked"] = ()
    for l in range(n_layers):
        p = f"gnn.layers.{l}"
        s[f"{p}.attn.merge.weight"] = (d, d, 1)
        s[f"{p}.attn.merge.bias"] = (d,)
        for k in range(3):
            s[f"{p}.attn.proj.{k}.weight"] = (d, d, 1)
            s[f"{p}.attn.proj.{k}.bias"] = (d,)
        s[f"{p}.mlp.0.weight"] = (2 * d, 2 * d, 1)
        s[f"{p}.mlp.0.bias"] = (2 * d,)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            s[f"{p}.mlp.1.{nm}"] = (2 * d,)
        s[f"{p}.mlp.1.num_batches_tracked"] = ()
        s[f"{p}.mlp.3.weight"] = (d, 2 * d, 1)
        s[f"{p}.mlp.3.bias"] = (d,)
    s["final_proj.weight"] = (d, d, 1)
    s["final_proj.bias"] = (d,)
    return s


# ---------------------------------------------------------------------------- weights
def _is_bn_key(key, shapes):
    stem = key.rsplit(".", 1)[0]
    return f"{stem}.running_mean" in shapes


def synth_state_dict(shapes, seed, gains=None):
    """Numpy state dict for `shapes`: conv weight ~N(0, 2/fan_in), conv bias ~N(0,0.05²),
    BN γ~U(0.5,1.5), β~N(0,0.1²), running_mean 0, running_var 1 (calibrate separately),
    num_batches_tracked 0 (int64).  `gains` maps a key substring to a weight multiplier."""
    gains = gains or {}
    sd = OrderedDict()
    for key, shape in shapes.items():
        n = int(np.prod(shape)) if len(shape) else 1
        leaf = key.rsplit(".", 1)[1] if "." in key else key
        if leaf == "num_batches_tracked":
            sd[key] = np.zeros((), dtype=np.int64)
            continue
        if key == "bin_score":
            sd[key] = np.float32(1.0).reshape(())
            continue
        if _is_bn_key(key, shapes):
            if leaf == "weight":
                v = np.float32(0.5) + uniform(seed, key, n)
            elif leaf == "bias":
                v = normal(seed, key, n) * np.float32(0.1)
            elif leaf == "running_mean":
                v = np.zeros(n, dtype=np.float32)
            else:
                v = np.ones(n, dtype=np.float32)
        elif leaf == "weight":
            fan_in = int(np.prod(shape[1:]))
            v = normal(seed, key, n) * np.float32(np.sqrt(np.float32(2.0 / fan_in)))
            for sub, g in gains.items():
                if sub in key:
                    v = v * np.float32(g)
        else:  # conv bias
            v = normal(seed, key, n) * np.float32(0.05)
        sd[key] = v.reshape(shape).astype(np.float32)
    return sd


def bn_stat_keys(shapes):
    return [k for k in shapes if k.endswith("running_mean") or k.endswith("running_var")]


def apply_bn_stats(sd, stats):
    """Overwrite running_mean / running_var entries with calibrated ones (dict of arrays)."""
    for k, v in stats.items():
        assert k in sd and sd[k].shape == np.asarray(v).shape, k
        sd[k] = np.asarray(v, dtype=np.float32)
    return sd


# ----------------------------------------------------------------------------- images
def _box5_wrap(a):
    out = np.zeros_like(a)
    for dy in (-2, -1, 0, 1, 2):
        row = np.roll(a, dy, axis=0)
        for dx in (-2, -1, 0, 1, 2):
            out = out + np.roll(row, dx, axis=1)
    return out * np.float32(1.0 / 25.0)


def synth_image(seed, H, W):
    """Smooth random texture in [0,1]: uniform noise, 5×5 box blur twice, min-max rescale."""
    a = uniform(seed, "image", H * W).reshape(H, W)
    a = _box5_wrap(_box5_wrap(a))
    lo, hi = a.min(), a.max()
    return ((a - lo) / (hi - lo)).astype(np.float32)


def synth_pair(seed, H, W, shift=(8, 16), noise=0.01):
    """(image0, image1) float32 (H,W): image1 = roll(image0, shift) + noise·N(0,1), clamped."""
    im0 = synth_image(seed, H, W)
    im1 = np.roll(im0, shift, axis=(0, 1)) + normal(seed, "image1_noise", H * W).reshape(H, W) * np.float32(noise)
    return im0, np.clip(im1, np.float32(0.0), np.float32(1.0)).astype(np.float32)


# ------------------------------------------------------------- canonical synthetic weight sets
# Shared by tests, bench.py and tests/golden/make_golden.py.  BatchNorm running statistics were
# calibrated once with the reference's own modules (make_golden.py) and are shipped as data.
SP_SEED, SG_SEED = 123, 456
SG_GAINS = {".mlp.3.weight": 0.3, "final_proj.weight": 0.7}
SG_CONFIGS = {      # descriptor_dim -> (keypoint_encoder, sinkhorn_iterations, match_threshold)
    64: ([32, 64], 30, 0.1),                  # README.md:134-140 pairing (descriptor_dim 64, HD = 16)
    128: ([32, 64, 128], 30, 0.1),            # superpoint_glue_test.py:23,33-35 defaults (C3)
    256: ([32, 64, 128, 256], 100, 0.2),      # SuperGlue.default_config (C5)
}
# Second SuperGlue weight set, "t" (round 4, VERDICT r3 task 1): the SAME seeded values with gains chosen for trained-model-like
# score statistics -- scores_in std = 5.0 on the calibration pair, bin_score = mean + 2 sigma (SURVEY 8c's recipe) -- and a tamer
# GNN (residual branch x0.1, q/k projections x0.5: attention logits a quarter as large).  The selection criterion was a property of
# the REFERENCE alone: on this set its own fp32 forward stays inside 1e-4 + 1e-4|x| of a float64 evaluation of itself on gnn17,
# scores_in and Z (worst element of three probe seeds: 5e-6 / 9e-5 / 1.1e-4 at |x| > 100), so the north_star bar is one a correct
# fp32 implementation can be held to element-wise, with no envelope clause.  The default set (SG_GAINS) is where it cannot: there
# the reference itself is 2e-4..2.6e-3 from its float64 self.  final_proj gains per descriptor_dim were solved for std = 5.0.
SGT_GAINS = {
    64: {".mlp.3.weight": 0.1, "attn.proj.0.weight": 0.5, "attn.proj.1.weight": 0.5, "final_proj.weight": 0.881},
    128: {".mlp.3.weight": 0.1, "attn.proj.0.weight": 0.5, "attn.proj.1.weight": 0.5, "final_proj.weight": 0.787},
    256: {".mlp.3.weight": 0.1, "attn.proj.0.weight": 0.5, "attn.proj.1.weight": 0.5, "final_proj.weight": 0.745},
}
_STATS = {}


def _stats(fname="synth_bn_stats.npz"):
    if fname not in _STATS:
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", fname)
        with np.load(path) as z:
            _STATS[fname] = {k: z[k] for k in z.files}
    return _STATS[fname]


def calibrated_stats(prefix):
    """prefix 'sp128' | 'sp256' | 'sg128' | 'sg256' -> {state-dict key: running stat array}."""
    return {k.split("/", 1)[1]: v for k, v in _stats().items()
            if k.startswith(prefix + "/") and not k.endswith("bin_score")}


def make_superpoint_state_dict(descriptor_dim=128, heavy=False):
    sd = synth_state_dict(superpoint_bn_shapes(descriptor_dim), SP_SEED)
    sd = apply_bn_stats(sd, calibrated_stats(f"sp{descriptor_dim}"))
    return heavy_superpoint(sd) if heavy else sd


# "Heavy" weight sets (round 5, VERDICT r4 item 3 / ADVICE r4): FUNCTION-PRESERVING re-parameterisations of the sets above by powers of
# two.  A BatchNorm channel's gamma and beta times 2^k scales that channel's activation by 2^k (eval mode; ReLU and max-pooling commute
# with a positive scale), and the next layer's weights of that input channel times 2^-k undo it: every product and every sum of the
# reference's fp32 forward is scaled EXACTLY, so the reference's outputs are bit-identical to the base set's (tests/golden/make_golden.py
# --heavy-check runs the reference on both and asserts it; tests/test_host.py does the same with the oracle) and every committed
# golden vector stays valid -- while the FOLDED weights an implementation sees become heavy-tailed: one output channel 2^10 above
# the rest (one column of a GNN layer's mlp.0' with a 2^10 times larger L1 norm: the bound that scales gnn_tail_h2's hidden
# activations gets 2^10 looser), one 2^-10 below, per-image maxima dominated by one channel, a query / key channel pair at 2^7 / 2^-7.
HEAVY_SP = (("down1.mpconv.1.conv", 0, 5, 10, "down1.mpconv.1.conv.3.weight"),       # conv2a channel 5 x 2^10, undone in conv2b
            ("down2.mpconv.1.conv", 0, 7, -10, "down2.mpconv.1.conv.3.weight"),      # conv3a channel 7 x 2^-10, undone in conv3b
            ("down3.mpconv.1.conv", 3, 11, 10, None))                                # conv4b channel 11 x 2^10, undone in convPa AND convDa
HEAVY_SG_LAYERS = {3: (9, 10), 11: (200, 10), 14: (77, -10)}                         # layer -> (hidden channel of mlp.0, exponent)
HEAVY_SG_QK = {5: (3, 7)}                                                            # layer -> (q/k channel, exponent): q x 2^k, k x 2^-k


def heavy_superpoint(sd):
    sd = OrderedDict((k, np.array(v, copy=True)) for k, v in sd.items())
    for stem, idx, ch, e, nxt in HEAVY_SP:
        f = np.float32(2.0 ** e)
        sd[f"{stem}.{idx + 1}.weight"][ch] *= f
        sd[f"{stem}.{idx + 1}.bias"][ch] *= f
        for key in ([nxt] if nxt else ["convPa.weight", "convDa.weight"]):
            sd[key][:, ch] *= np.float32(2.0 ** -e)
    return sd


def heavy_superglue(sd):
    sd = OrderedDict((k, np.array(v, copy=True)) for k, v in sd.items())
    for l, (ch, e) in HEAVY_SG_LAYERS.items():
        f = np.float32(2.0 ** e)
        sd[f"gnn.layers.{l}.mlp.1.weight"][ch] *= f
        sd[f"gnn.layers.{l}.mlp.1.bias"][ch] *= f
        sd[f"gnn.layers.{l}.mlp.3.weight"][:, ch] *= np.float32(2.0 ** -e)
    for l, (ch, e) in HEAVY_SG_QK.items():
        for k, ee in ((0, e), (1, -e)):
            sd[f"gnn.layers.{l}.attn.proj.{k}.weight"][ch] *= np.float32(2.0 ** ee)
            sd[f"gnn.layers.{l}.attn.proj.{k}.bias"][ch] *= np.float32(2.0 ** ee)
    return sd


def make_superglue_state_dict(descriptor_dim=128, keypoint_encoder=None, n_layers=18, variant="default", heavy=False):
    if heavy:
        return heavy_superglue(make_superglue_state_dict(descriptor_dim, keypoint_encoder, n_layers, variant))
    return _make_superglue_state_dict(descriptor_dim, keypoint_encoder, n_layers, variant)


def _make_superglue_state_dict(descriptor_dim=128, keypoint_encoder=None, n_layers=18, variant="default"):
    """variant "default": SG_GAINS (heavy-tailed scores, |S| up to several hundred); "t": SGT_GAINS (trained-model-like score
    statistics, see above).  Both carry BatchNorm running statistics calibrated with the reference's own modules."""
    kenc = list(keypoint_encoder) if keypoint_encoder is not None else SG_CONFIGS[descriptor_dim][0]
    if variant == "t":
        sd = synth_state_dict(superglue_shapes(descriptor_dim, kenc, n_layers), SG_SEED, gains=SGT_GAINS[descriptor_dim])
        st = _stats("synth_bn_stats_t.npz")
        pre = f"sgt{descriptor_dim}/"
        apply_bn_stats(sd, {k[len(pre):]: v for k, v in st.items() if k.startswith(pre) and not k.endswith("bin_score") and k[len(pre):] in sd})
        sd["bin_score"] = np.float32(st[pre + "bin_score"]).reshape(())
        return sd
    assert variant == "default", variant
    sd = synth_state_dict(superglue_shapes(descriptor_dim, kenc, n_layers), SG_SEED, gains=SG_GAINS)
    apply_bn_stats(sd, {k: v for k, v in calibrated_stats(f"sg{descriptor_dim}").items() if k in sd})
    sd["bin_score"] = np.float32(_stats()[f"sg{descriptor_dim}/bin_score"]).reshape(())
    return sd

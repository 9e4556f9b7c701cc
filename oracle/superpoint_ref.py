"""TEST INFRASTRUCTURE (checker only) — CPU restatement of the reference SuperPoint forward.

Follows superpoint/models/superpoint_test.py (BN variant) and superglue/models/superpoint.py
(official variant, no BN) of the reference, written as plain functions over a state dict
(torch CPU tensor ops, fp32).  Line citations are relative to /root/reference.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, unet_parts.py:16,19; superpoint_test.py:77-84


def _t(sd, key):
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.as_tensor(v)


def _bn(x, sd, prefix):
    # eval-mode BatchNorm with running statistics (superpoint_glue_test.py:69 `.eval()`)
    return F.batch_norm(x, _t(sd, prefix + ".running_mean"), _t(sd, prefix + ".running_var"),
                        _t(sd, prefix + ".weight"), _t(sd, prefix + ".bias"), False, 0.0, BN_EPS)


def _double_conv(x, sd, prefix):
    # unet_parts.py:10-25: (conv3x3 pad1 -> BN -> ReLU) x 2
    for idx in (0, 3):
        x = F.conv2d(x, _t(sd, f"{prefix}.{idx}.weight"), _t(sd, f"{prefix}.{idx}.bias"), padding=1)
        x = F.relu(_bn(x, sd, f"{prefix}.{idx + 1}"))
    return x


def encoder_bn(x, sd):
    """superpoint_test.py:113-116 -> x4 (B,128,H/8,W/8)."""
    x1 = _double_conv(x, sd, "inc.conv.conv")                               # :113, unet_parts.py:28-35
    x2 = _double_conv(F.max_pool2d(x1, 2), sd, "down1.mpconv.1.conv")       # :114, unet_parts.py:38-48
    x3 = _double_conv(F.max_pool2d(x2, 2), sd, "down2.mpconv.1.conv")       # :115
    x4 = _double_conv(F.max_pool2d(x3, 2), sd, "down3.mpconv.1.conv")       # :116
    return x4


def heads_bn(x4, sd):
    """superpoint_test.py:119-126 -> semi (B,65,h,w), desc (B,d,h,w) divided by its channel norm."""
    cPa = F.relu(_bn(F.conv2d(x4, _t(sd, "convPa.weight"), _t(sd, "convPa.bias"), padding=1), sd, "bnPa"))
    semi = _bn(F.conv2d(cPa, _t(sd, "convPb.weight"), _t(sd, "convPb.bias")), sd, "bnPb")
    cDa = F.relu(_bn(F.conv2d(x4, _t(sd, "convDa.weight"), _t(sd, "convDa.bias"), padding=1), sd, "bnDa"))
    desc = _bn(F.conv2d(cDa, _t(sd, "convDb.weight"), _t(sd, "convDb.bias")), sd, "bnDb")
    dn = torch.norm(desc, p=2, dim=1)                                       # :125 (no eps)
    desc = desc.div(torch.unsqueeze(dn, 1))                                 # :126
    return semi, desc


def encoder_official(x, sd):
    """superglue/models/superpoint.py:147-158."""
    r = F.relu
    x = r(F.conv2d(x, _t(sd, "conv1a.weight"), _t(sd, "conv1a.bias"), padding=1))
    x = r(F.conv2d(x, _t(sd, "conv1b.weight"), _t(sd, "conv1b.bias"), padding=1))
    x = F.max_pool2d(x, 2, 2)
    x = r(F.conv2d(x, _t(sd, "conv2a.weight"), _t(sd, "conv2a.bias"), padding=1))
    x = r(F.conv2d(x, _t(sd, "conv2b.weight"), _t(sd, "conv2b.bias"), padding=1))
    x = F.max_pool2d(x, 2, 2)
    x = r(F.conv2d(x, _t(sd, "conv3a.weight"), _t(sd, "conv3a.bias"), padding=1))
    x = r(F.conv2d(x, _t(sd, "conv3b.weight"), _t(sd, "conv3b.bias"), padding=1))
    x = F.max_pool2d(x, 2, 2)
    x = r(F.conv2d(x, _t(sd, "conv4a.weight"), _t(sd, "conv4a.bias"), padding=1))
    x = r(F.conv2d(x, _t(sd, "conv4b.weight"), _t(sd, "conv4b.bias"), padding=1))
    return x


def heads_official(x4, sd):
    """superglue/models/superpoint.py:161-162,190-192: semi, L2-normalised dense descriptors."""
    cPa = F.relu(F.conv2d(x4, _t(sd, "convPa.weight"), _t(sd, "convPa.bias"), padding=1))
    semi = F.conv2d(cPa, _t(sd, "convPb.weight"), _t(sd, "convPb.bias"))
    cDa = F.relu(F.conv2d(x4, _t(sd, "convDa.weight"), _t(sd, "convDa.bias"), padding=1))
    desc = F.conv2d(cDa, _t(sd, "convDb.weight"), _t(sd, "convDb.bias"))
    desc = F.normalize(desc, p=2, dim=1)                                    # :192 (eps 1e-12)
    return semi, desc


def score_map(semi):
    """superpoint_test.py:128-131: softmax over 65 channels, drop dustbin, 8x8 pixel shuffle."""
    scores = F.softmax(semi, 1)[:, :-1]
    b, _, h, w = scores.shape
    scores = scores.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8)
    return scores.permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)


def simple_nms(scores, nms_radius):
    """superpoint_test.py:7-22 (scores: (B,H,W)); max_pool on a 4-D view — identical values."""
    assert nms_radius >= 0
    k = nms_radius * 2 + 1

    def max_pool(x):
        return F.max_pool2d(x[:, None], kernel_size=k, stride=1, padding=nms_radius)[:, 0]

    zeros = torch.zeros_like(scores)
    max_mask = scores == max_pool(scores)
    for _ in range(2):
        supp_mask = max_pool(max_mask.float()) > 0
        supp_scores = torch.where(supp_mask, zeros, scores)
        new_max_mask = supp_scores == max_pool(supp_scores)
        max_mask = max_mask | (new_max_mask & (~supp_mask))
    return torch.where(max_mask, scores, zeros)


def extract_keypoints(nms_scores, threshold, border, max_keypoints):
    """superpoint_test.py:135-151 for ONE image (H8,W8): -> keypoints (K,2) float (x,y), scores (K,).

    nonzero is row-major (y,x); remove_borders :25-30; top_k :33-37 (torch.topk, sorted desc;
    ties resolved here as the reference's torch build does); flip to (x,y) :151."""
    H8, W8 = nms_scores.shape
    kp = torch.nonzero(nms_scores > threshold)
    sc = nms_scores[tuple(kp.t())]
    mask = (kp[:, 0] >= border) & (kp[:, 0] < H8 - border) & (kp[:, 1] >= border) & (kp[:, 1] < W8 - border)
    kp, sc = kp[mask], sc[mask]
    if max_keypoints >= 0 and max_keypoints < len(kp):
        sc, idx = torch.topk(sc, max_keypoints, dim=0)
        kp = kp[idx]
    return torch.flip(kp, [1]).float(), sc


def sample_descriptors(keypoints, descriptors, s=8, align_corners=False):
    """superpoint_test.py:40-52.  keypoints (1,K,2) (x,y) px; descriptors (1,C,h,w).
    `align_corners` is what `int(torch.__version__[2]) > 2` (:47) evaluates to: False under
    torch>=1.10/2.x (incl. the 2.10 used here), True under torch 1.3-1.9."""
    b, c, h, w = descriptors.shape
    keypoints = keypoints - s / 2 + 0.5
    keypoints = keypoints / torch.tensor([(w * s - s / 2 - 0.5), (h * s - s / 2 - 0.5)]).to(keypoints)[None]
    keypoints = keypoints * 2 - 1
    d = F.grid_sample(descriptors, keypoints.view(b, 1, -1, 2), mode="bilinear",
                      align_corners=align_corners)
    return F.normalize(d.reshape(b, c, -1), p=2, dim=1)


def superpoint_forward(x, sd, config, variant="bn", align_corners=False, return_dense=False):
    """Full forward: superpoint_test.py:103-161 (variant 'bn') / superpoint.py:145-202 ('official').
    x (B,1,H,W) float32.  Returns the reference's dict of lists (+ dense intermediates)."""
    cfg = {"descriptor_dim": 256, "nms_radius": 4, "keypoint_threshold": 0.005,
           "max_keypoints": -1, "remove_borders": 4}
    cfg.update(config)
    with torch.no_grad():
        if variant == "bn":
            x4 = encoder_bn(x, sd)
            semi, desc = heads_bn(x4, sd)
        else:
            x4 = encoder_official(x, sd)
            semi, desc = heads_official(x4, sd)
        smap = score_map(semi)
        nms = simple_nms(smap, cfg["nms_radius"])
        kpts, scs, descs = [], [], []
        for b in range(x.shape[0]):
            k, s = extract_keypoints(nms[b], cfg["keypoint_threshold"], cfg["remove_borders"],
                                     cfg["max_keypoints"])
            kpts.append(k)
            scs.append(s)
            descs.append(sample_descriptors(k[None], desc[b][None], 8, align_corners)[0])
    out = {"keypoints": kpts, "scores": tuple(scs), "descriptors": descs}
    if return_dense:
        out.update({"x4": x4, "semi": semi, "desc": desc, "score_map": smap, "nms": nms})
    return out

"""TEST INFRASTRUCTURE (checker only) — CPU restatement of the reference SuperGlue forward.

Follows superglue/models/superglue_test.py of the reference as plain functions over a
state dict (torch CPU tensor ops, fp32).  Line citations are relative to /root/reference.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _t(sd, key):
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.as_tensor(v)


def _conv1(x, sd, prefix):
    return F.conv1d(x, _t(sd, prefix + ".weight"), _t(sd, prefix + ".bias"))


def _bn(x, sd, prefix):
    return F.batch_norm(x, _t(sd, prefix + ".running_mean"), _t(sd, prefix + ".running_var"),
                        _t(sd, prefix + ".weight"), _t(sd, prefix + ".bias"), False, 0.0, BN_EPS)


def _mlp(x, sd, prefix, n_convs):
    """superglue_test.py:49-60: conv1d(k=1) [+BN+ReLU after all but the last]."""
    for i in range(n_convs):
        j = 3 * i
        x = _conv1(x, sd, f"{prefix}.{j}")
        if i < n_convs - 1:
            x = F.relu(_bn(x, sd, f"{prefix}.{j + 1}"))
    return x


def normalize_keypoints(kpts, image_shape):
    """superglue_test.py:63-70."""
    _, _, height, width = image_shape
    one = kpts.new_tensor(1)
    size = torch.stack([one * width, one * height])[None]
    center = size / 2
    scaling = size.max(1, keepdim=True).values * 0.7
    return (kpts - center[:, None, :]) / scaling[:, None, :]


def keypoint_encoder(kpts, scores, sd, n_convs):
    """superglue_test.py:73-82."""
    inputs = [kpts.transpose(1, 2), scores.unsqueeze(1)]
    return _mlp(torch.cat(inputs, dim=1), sd, "kenc.encoder", n_convs)


def attention(query, key, value):
    """superglue_test.py:85-89."""
    dim = query.shape[1]
    scores = torch.einsum("bdhn,bdhm->bhnm", query, key) / dim ** .5
    prob = F.softmax(scores, dim=-1)
    return torch.einsum("bhnm,bdhm->bdhn", prob, value)


def propagation(x, source, sd, prefix, num_heads=4):
    """superglue_test.py:92-119: multi-head attention (dim-major/head-minor view :104) + MLP."""
    b, d, _ = x.shape
    dim = d // num_heads
    q = _conv1(x, sd, f"{prefix}.attn.proj.0").view(b, dim, num_heads, -1)
    k = _conv1(source, sd, f"{prefix}.attn.proj.1").view(b, dim, num_heads, -1)
    v = _conv1(source, sd, f"{prefix}.attn.proj.2").view(b, dim, num_heads, -1)
    msg = attention(q, k, v).contiguous().view(b, d, -1)
    msg = _conv1(msg, sd, f"{prefix}.attn.merge")
    return _mlp(torch.cat([x, msg], dim=1), sd, f"{prefix}.mlp", 2)


def gnn(desc0, desc1, sd, layer_names, taps=None):
    """superglue_test.py:122-138.  `taps`: optional dict filled with desc0 after listed layers."""
    for i, name in enumerate(layer_names):
        if name == "cross":
            src0, src1 = desc1, desc0
        else:
            src0, src1 = desc0, desc1
        p = f"gnn.layers.{i}"
        delta0, delta1 = propagation(desc0, src0, sd, p), propagation(desc1, src1, sd, p)
        desc0, desc1 = desc0 + delta0, desc1 + delta1
        if taps is not None:
            taps[i] = (desc0.clone(), desc1.clone())
    return desc0, desc1


def log_sinkhorn_iterations(Z, log_mu, log_nu, iters):
    """superglue_test.py:141-147."""
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1)


def log_optimal_transport(scores, alpha, iters):
    """superglue_test.py:150-170."""
    b, m, n = scores.shape
    one = scores.new_tensor(1)
    ms, ns = (m * one).to(scores), (n * one).to(scores)
    bins0 = alpha.expand(b, m, 1)
    bins1 = alpha.expand(b, 1, n)
    alpha = alpha.expand(b, 1, 1)
    couplings = torch.cat([torch.cat([scores, bins0], -1), torch.cat([bins1, alpha], -1)], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])
    log_mu, log_nu = log_mu[None].expand(b, -1), log_nu[None].expand(b, -1)
    Z = log_sinkhorn_iterations(couplings, log_mu, log_nu, iters)
    return Z - norm


def extract_matches(scores, match_threshold):
    """superglue_test.py:268-285 (scores = Z (B,M+1,N+1))."""
    max0, max1 = scores[:, :-1, :-1].max(2), scores[:, :-1, :-1].max(1)
    indices0, indices1 = max0.indices, max1.indices
    ar0 = torch.arange(indices0.shape[1])[None]
    ar1 = torch.arange(indices1.shape[1])[None]
    mutual0 = ar0 == indices1.gather(1, indices0)
    mutual1 = ar1 == indices0.gather(1, indices1)
    zero = scores.new_tensor(0)
    mscores0 = torch.where(mutual0, max0.values.exp(), zero)
    mscores1 = torch.where(mutual1, mscores0.gather(1, indices1), zero)
    valid0 = mutual0 & (mscores0 > match_threshold)
    valid1 = mutual1 & valid0.gather(1, indices1)
    indices0 = torch.where(valid0, indices0, indices0.new_tensor(-1))
    indices1 = torch.where(valid1, indices1, indices1.new_tensor(-1))
    return indices0, indices1, mscores0, mscores1


def superglue_forward(data, sd, config, return_dense=False):
    """superglue_test.py:230-285.  data: descriptors0/1 (B,d,N), keypoints0/1 (B,N,2),
    scores0/1 (B,N), image0/1 (shape only, or pass 'image_shape0/1' tuples)."""
    cfg = {"descriptor_dim": 256, "keypoint_encoder": [32, 64, 128, 256],
           "GNN_layers": ["self", "cross"] * 9, "sinkhorn_iterations": 100, "match_threshold": 0.2}
    cfg.update(config)
    with torch.no_grad():
        desc0, desc1 = data["descriptors0"], data["descriptors1"]
        kpts0, kpts1 = data["keypoints0"], data["keypoints1"]
        if kpts0.shape[1] == 0 or kpts1.shape[1] == 0:                      # :235-242
            shape0, shape1 = kpts0.shape[:-1], kpts1.shape[:-1]
            return {"matches0": kpts0.new_full(shape0, -1, dtype=torch.int),
                    "matches1": kpts1.new_full(shape1, -1, dtype=torch.int),
                    "matching_scores0": kpts0.new_zeros(shape0),
                    "matching_scores1": kpts1.new_zeros(shape1)}
        shp0 = data["image0"].shape if "image0" in data else data["image_shape0"]
        shp1 = data["image1"].shape if "image1" in data else data["image_shape1"]
        kpts0 = normalize_keypoints(kpts0, shp0)
        kpts1 = normalize_keypoints(kpts1, shp1)
        n_convs = len(cfg["keypoint_encoder"]) + 1
        desc0 = desc0 + keypoint_encoder(kpts0, data["scores0"], sd, n_convs)  # :249
        desc1 = desc1 + keypoint_encoder(kpts1, data["scores1"], sd, n_convs)  # :250
        dense = {"kenc0": desc0.clone(), "kenc1": desc1.clone()}
        taps = {} if return_dense else None
        desc0, desc1 = gnn(desc0, desc1, sd, cfg["GNN_layers"], taps)
        mdesc0, mdesc1 = _conv1(desc0, sd, "final_proj"), _conv1(desc1, sd, "final_proj")
        scores = torch.einsum("bdn,bdm->bnm", mdesc0, mdesc1)
        scores = scores / cfg["descriptor_dim"] ** .5
        Z = log_optimal_transport(scores, _t(sd, "bin_score"), iters=cfg["sinkhorn_iterations"])
        i0, i1, ms0, ms1 = extract_matches(Z, cfg["match_threshold"])
    out = {"matches0": i0, "matches1": i1, "matching_scores0": ms0, "matching_scores1": ms1}
    if return_dense:
        dense.update({"gnn_taps": taps, "gnn0": desc0, "gnn1": desc1, "mdesc0": mdesc0,
                      "mdesc1": mdesc1, "scores_in": scores, "Z": Z})
        out["dense"] = dense
    return out

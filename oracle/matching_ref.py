"""TEST INFRASTRUCTURE (checker only) — restatement of superglue/models/matching_test.py:54-82."""
import torch

from .superglue_ref import superglue_forward
from .superpoint_ref import superpoint_forward


def matching_forward(data, sd_sp, sd_sg, config, variant="bn", align_corners=False):
    pred = {}
    sp_cfg, sg_cfg = config.get("superpoint", {}), config.get("superglue", {})
    if "keypoints0" not in data:                                            # :63-65
        p0 = superpoint_forward(data["image0"], sd_sp, sp_cfg, variant, align_corners)
        pred.update({k + "0": v for k, v in p0.items()})
    if "keypoints1" not in data:                                            # :66-68
        p1 = superpoint_forward(data["image1"], sd_sp, sp_cfg, variant, align_corners)
        pred.update({k + "1": v for k, v in p1.items()})
    data = {**data, **pred}                                                 # :73
    for k in data:                                                          # :75-77
        if isinstance(data[k], (list, tuple)):
            data[k] = torch.stack(data[k])
    return {**pred, **superglue_forward(data, sd_sg, sg_cfg)}               # :80

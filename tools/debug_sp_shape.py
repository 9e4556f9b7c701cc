#!/usr/bin/env python
"""Developer tool: which SuperPoint stage first departs from the oracle for a given image size?  (GPU box)
usage: python tools/debug_sp_shape.py H W [B]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import superpoint_ref as R  # noqa: E402
from tests import util  # noqa: E402
from image_matching_amd import _lib as L  # noqa: E402
from image_matching_amd.engine import Engine  # noqa: E402

H, W = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sd = util.sp_sd(128)
x = torch.cat([util.pair(900 + b, H, W)[0] for b in range(B)])
with torch.no_grad():
    x1 = R._double_conv(x, sd, "inc.conv.conv")
    x2 = R._double_conv(F.max_pool2d(x1, 2), sd, "down1.mpconv.1.conv")
    x3 = R._double_conv(F.max_pool2d(x2, 2), sd, "down2.mpconv.1.conv")
    x4 = R._double_conv(F.max_pool2d(x3, 2), sd, "down3.mpconv.1.conv")
    semi, desc = R.heads_bn(x4, sd)
ref = {"a1": F.max_pool2d(x1, 2), "a2": F.max_pool2d(x2, 2), "a3": F.max_pool2d(x3, 2), "x4": x4, "semi": semi}
for conv in (os.environ.get("IMX_CONV", "wino"),):
    eng = Engine(util.sp_config(128, 100), util.sg_config(128), "cuda")
    eng.load_state_dict(L.NET_SUPERPOINT, sd)
    eng.set_debug(True)
    eng.superpoint(x.cuda())
    for k, r in ref.items():
        got = np.transpose(eng.fetch(k), (0, 3, 1, 2))
        r = r.numpy()
        err = np.abs(got - r)
        bad = err > 1e-4 + 1e-4 * np.abs(r)
        msg = "%s %s: max err %.3e, %d / %d out of tolerance" % (conv, k, err.max(), bad.sum(), bad.size)
        if bad.any():
            idx = np.argwhere(bad)
            msg += "; bad b %s c %s..%s y %s..%s x %s..%s" % (sorted(set(idx[:, 0].tolist())), idx[:, 1].min(), idx[:, 1].max(), idx[:, 2].min(), idx[:, 2].max(),
                                                        idx[:, 3].min(), idx[:, 3].max())
            ys = sorted(set(idx[:, 2].tolist()))
            xs = sorted(set(idx[:, 3].tolist()))
            msg += "\n     rows %s\n     cols %s" % (ys[:40], xs[:40])
        print(msg)

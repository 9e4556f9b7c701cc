import sys, torch, numpy as np
sys.path.insert(0,'/root/repo')
from tests import util
from image_matching_amd import _lib as L, synth
from image_matching_amd.engine import Engine
eng = Engine(util.sp_config(128, 1024), util.sg_config(128), "cuda")
eng.load_state_dict(L.NET_SUPERPOINT, util.sp_sd(128))
def run(xs, tag):
    for mode in ("wino", "wino_h"):
        eng.set_option("conv", mode)
        for _ in range(2): eng.superpoint_dense(xs)
        eng.timing_reset(); eng.set_timing(True)
        for _ in range(6): eng.superpoint_dense(xs)
        torch.cuda.synchronize()
        rows = {r[0]: (r[2]/r[1], r[3]) for r in eng.timing_report(forms=True)}
        eng.set_timing(False)
        print(tag, mode, "conv1ab %.3f ms %s | conv2a %.3f" % (rows["conv1ab_pool"][0], rows["conv1ab_pool"][1], rows["conv2a"][0]))
ims = [synth.synth_pair(i, 480, 640) for i in range(64)]
xs = torch.from_numpy(np.stack([p[k] for p in ims for k in (0,1)]))[:, None].cuda()
run(xs, "synthetic pairs")
run(torch.rand(128,1,480,640, device="cuda"), "uniform noise  ")
run(torch.zeros(128,1,480,640, device="cuda"), "zeros          ")

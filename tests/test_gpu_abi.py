"""The C ABI used from plain C on the GPU: examples/abi_example.c compiled as C99 against include/imx.h, run to
completion (weights from a record file, one pair through imx_match_pairs on its own stream) and compared with the
ctypes path on the same inputs.  Also: repeated load_state_dict does not grow the handle's HBM footprint."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _write_records(path, nets):
    with open(path, "wb") as f:
        for net, sd in nets:
            for key, val in sd.items():
                if key.endswith("num_batches_tracked"):
                    continue
                a = np.ascontiguousarray(val.numpy(), dtype=np.float32)
                kb = key.encode()
                f.write(struct.pack("<ii", net, len(kb)) + kb + struct.pack("<i", a.ndim))
                f.write(struct.pack(f"<{a.ndim}q", *a.shape))
                f.write(a.tobytes())


def test_plain_c_example_runs_matching_forward_and_agrees_with_ctypes(tmp_path):
    from image_matching_amd import _lib
    from image_matching_amd.superglue.models.matching_test import Matching
    if not shutil.which("gcc"):
        pytest.skip("gcc not present")
    d, K, H, W, seed = 128, 300, 240, 320, 21
    sd_sp, sd_sg = util.sp_sd(d), util.sg_sd(d)
    _write_records(tmp_path / "weights.bin", [(0, sd_sp), (1, sd_sg)])
    x0, x1 = util.pair(seed, H, W)
    with open(tmp_path / "pair.bin", "wb") as f:
        f.write(x0.numpy().astype(np.float32).tobytes() + x1.numpy().astype(np.float32).tobytes())
    exe, libdir = str(tmp_path / "abi_example"), os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                    "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "examples", "abi_example.c"), "-L" + libdir, "-limx",
                    "-L/opt/rocm/lib", "-lamdhip64", "-o", exe], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(tmp_path / "weights.bin"), str(tmp_path / "pair.bin"), str(H), str(W), str(K)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    # the same pair through the Python drop-in (ctypes)
    m = Matching({"superpoint": util.sp_config(d, K), "superglue": util.sg_config(d)}).eval().to("cuda")
    m.superpoint.load_state_dict(sd_sp)
    m.superglue.load_state_dict(sd_sg)
    out = m.match_batch(x0.cuda(), x1.cuda())
    m0 = out["matches0"][0].cpu().numpy()
    cs = 0
    for i, v in enumerate(m0):
        cs = (cs + (i + 1) * (int(v) + 2)) & 0x7fffffff
    lines = r.stdout.splitlines()
    assert f"keypoints {int(out['counts0'][0])} {int(out['counts1'][0])}" in lines, r.stdout
    assert f"matches {int((m0 >= 0).sum())} checksum {cs}" in lines, r.stdout
    assert int((m0 >= 0).sum()) > 0


def test_reloading_weights_does_not_leak_device_memory():
    """imx_finalize_weights frees the previous uploads of the net: a checkpoint sweep keeps a flat footprint."""
    from image_matching_amd import _lib as L
    from image_matching_amd.engine import Engine
    eng = Engine(util.sp_config(128, 64), util.sg_config(128), "cuda")
    sd_sp, sd_sg = util.sp_sd(128), util.sg_sd(128)
    eng.load_state_dict(L.NET_SUPERPOINT, sd_sp)
    eng.load_state_dict(L.NET_SUPERGLUE, sd_sg)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(6):
        eng.load_state_dict(L.NET_SUPERPOINT, sd_sp)
        eng.load_state_dict(L.NET_SUPERGLUE, sd_sg)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 6 reloads (one set of weights is ~40 MiB)"
    x = util.pair(3, 120, 160)[0].cuda()
    kp, sc, ds, n = eng.superpoint(x)            # and the reloaded weights work
    assert n[0] > 0


def test_gather_records_through_the_c_abi_over_rccl():
    """imx_gather_records: the path's one collective for hosts without torch.distributed.  A communicator of one rank is created
    with RCCL directly (ctypes: ncclGetUniqueId / ncclCommInitRank -- what a C host does), the records of a small batch are packed by
    imx_pack_records and gathered to rank 0 through the C ABI; the result equals the buffer that went in."""
    import ctypes
    import glob
    import os
    import torch
    from image_matching_amd import _lib as L, shard
    from image_matching_amd.superglue.models.matching_test import Matching
    from tests import util
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["/opt/rocm/lib/librccl.so"]
    rccl = next((ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL) for c in cands if os.path.exists(c)), None)
    assert rccl is not None, "no RCCL library found"

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid, comm = UniqueId(), ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        d, K, H, W = 128, 256, 240, 320
        m = Matching({"superpoint": util.sp_config(d, K), "superglue": util.sg_config(d)}).eval().to("cuda")
        m.superpoint.load_state_dict(util.sp_sd(d))
        m.superglue.load_state_dict(util.sg_sd(d))
        pairs = [util.pair(60 + i, H, W) for i in range(2)]
        out = m.match_batch(torch.cat([p[0] for p in pairs]).cuda(), torch.cat([p[1] for p in pairs]).cuda())
        rec = m.pack_records([5, 9], out, pad_to=3)
        eng = m._shared.get_engine([0, 1])
        got = torch.full_like(rec, -7)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = eng.lib.imx_gather_records(eng.handle, ctypes.c_void_p(rec.data_ptr()), rec.shape[0], rec.shape[1],
                                        ctypes.c_void_p(got.data_ptr()), 0, comm, st)
        assert rc == 0, eng.lib.imx_last_error(eng.handle).decode()
        torch.cuda.synchronize()
        assert torch.equal(got, rec) and shard.unpack_records(got)["pair_id"].tolist() == [5, 9]
        assert eng.lib.imx_gather_records(eng.handle, ctypes.c_void_p(rec.data_ptr()), 3, rec.shape[1], None, 0, comm, st) != 0   # dst without a buffer
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)

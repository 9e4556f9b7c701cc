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


def test_bad_calls_return_errors_with_messages_and_leave_the_handle_usable():
    """Every entry point returns 0 / <0 and never throws or aborts across the ABI (include/imx.h).  A sequence of wrong calls --
    forward before the weights are loaded, a state-dict key the reference's module does not have, a tensor of the wrong shape
    (the reference's load_state_dict messages: superpoint_test.py:87-99), a missing key at finalize, shapes that make no sense,
    unknown options and taps, null pointers -- each fails with a message, and the same handle then loads the weights and matches
    a pair as if nothing had happened."""
    import ctypes
    from image_matching_amd import _lib as L
    from image_matching_amd.engine import Engine, ImxError
    d, K, H, W = 128, 200, 120, 160
    eng = Engine(util.sp_config(d, K), util.sg_config(d), "cuda")
    lib, h = eng.lib, eng.handle
    err = lambda: lib.imx_last_error(h).decode()
    x = torch.cat(util.pair(3, H, W)).cuda()
    counts = torch.zeros(2, dtype=torch.int32, device="cuda")
    fp = lambda t: ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))
    # forward before load_state_dict
    assert lib.imx_superpoint_detect(h, fp(x), 2, H, W, counts.data_ptr(), None) < 0 and "not finalized" in err()
    # keys and shapes
    w = torch.zeros(64, 1, 3, 3)
    shape = (ctypes.c_int64 * 4)(64, 1, 3, 3)
    assert lib.imx_load_weight(h, L.NET_SUPERPOINT, b"inc.conv.conv.0.weights", w.data_ptr(), 4, shape) < 0 and "unexpected key" in err().lower()
    bad = (ctypes.c_int64 * 4)(32, 1, 3, 3)
    assert lib.imx_load_weight(h, L.NET_SUPERPOINT, b"inc.conv.conv.0.weight", w.data_ptr(), 4, bad) < 0 and "size mismatch" in err()
    assert lib.imx_load_weight(h, 7, b"inc.conv.conv.0.weight", w.data_ptr(), 4, shape) < 0 and "net" in err()
    assert lib.imx_load_weight(h, L.NET_SUPERPOINT, None, w.data_ptr(), 4, shape) < 0
    assert lib.imx_load_weight(h, L.NET_SUPERPOINT, b"inc.conv.conv.0.weight", w.data_ptr(), 4, shape) == 0
    assert lib.imx_finalize_weights(h, L.NET_SUPERPOINT) < 0 and "Missing key" in err()
    # options, taps
    assert lib.imx_set_option(h, b"mfma", b"fp8") < 0 and lib.imx_set_option(h, b"nonsense", b"1") < 0 and lib.imx_set_option(h, None, b"1") < 0
    assert lib.imx_get_option(h, b"nonsense") == b""
    with pytest.raises(ImxError):
        eng.fetch("no_such_tap")
    # now the real weights; then wrong shapes on the forward calls
    eng.load_state_dict(L.NET_SUPERPOINT, util.sp_sd(d))
    eng.load_state_dict(L.NET_SUPERGLUE, util.sg_sd(d))
    assert lib.imx_superpoint_detect(h, fp(x), 0, H, W, counts.data_ptr(), None) < 0 and "shape" in err()
    assert lib.imx_superpoint_detect(h, fp(x), 2, 4, W, counts.data_ptr(), None) < 0
    kp = torch.zeros(1, 50, 2, device="cuda"); sc = torch.zeros(1, 50, device="cuda"); ds = torch.zeros(1, d, 50, device="cuda")
    m = torch.zeros(1, 50, dtype=torch.int64, device="cuda"); ms = torch.zeros(1, 50, device="cuda")
    args = lambda B, N0: (h, B, fp(kp), fp(sc), fp(ds), d * 50, 50, 1, None, N0, H, W, fp(kp), fp(sc), fp(ds), d * 50, 50, 1, None, 50, H, W,
                          m.data_ptr(), m.data_ptr(), fp(ms), fp(ms), None)
    assert lib.imx_superglue_forward(*args(0, 50)) < 0 and "shape" in err()
    assert lib.imx_superglue_forward(*args(1, -3)) < 0
    with pytest.raises(ImxError):                          # descriptor width that is not the configured one
        eng.superglue(kp, sc, torch.zeros(1, 64, 50, device="cuda"), (1, 1, H, W), kp, sc, ds, (1, 1, H, W))
    assert lib.imx_estimate_affine_partial(h, fp(kp), fp(kp), m.data_ptr(), None, 0, 50, 7.0, 16, 1, fp(ms), m.data_ptr(), counts.data_ptr(), None) < 0
    assert lib.imx_timing_report(h, 10 ** 6, None, None, None) < 0
    # and the handle works
    out = eng.match_pairs(x[:1], x[1:], want_desc=True)
    torch.cuda.synchronize()
    assert int(out["counts0"][0]) == K and int((out["matches0"] >= 0).sum()) > 0

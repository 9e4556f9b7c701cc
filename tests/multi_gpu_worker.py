"""One rank of tests/test_gpu_multi.py (launched by `python -m torch.distributed.run`, one process per GPU, backend nccl = RCCL):
pair-sharded matching + the path's one collective, both ways --
  * shard.gather_records (torch.distributed gather over RCCL),
  * imx_gather_records (the C-ABI form: an RCCL communicator made with ncclCommInitRank, as a host without torch would),
and, on rank 0, the comparison of the gathered record bytes with the same pairs matched as ONE batch on one GPU.
SURVEY section 8(e); reference: pairs are independent (superpoint_glue_test.py:66,72-78).  Prints 'MULTI-GPU OK ...' on rank 0."""
import ctypes
import glob
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_matching_amd import shard  # noqa: E402
from image_matching_amd.superglue.models.matching_test import Matching  # noqa: E402
from tests import util  # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
    n_pairs = int(os.environ.get("IMX_MULTI_PAIRS", "8"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    d, K, H, W = 128, 256, 240, 320

    def build():
        m = Matching({"superpoint": util.sp_config(d, K), "superglue": util.sg_config(d)}).eval().to(dev)
        m.superpoint.load_state_dict(util.sp_sd(d))
        m.superglue.load_state_dict(util.sg_sd(d))
        # results must not depend on how many pairs share a call: the throughput forms for every batch size
        m._shared.get_engine([0, 1]).set_option("latency_forms", "off")
        return m

    def run(m, ids):
        pairs = [util.pair(60 + i, H, W) for i in ids]
        return m.match_batch(torch.cat([p[0] for p in pairs]).to(dev), torch.cat([p[1] for p in pairs]).to(dev))

    m = build()
    mine = shard.shard_indices(n_pairs, rank, world)
    rows = shard.shard_rows(n_pairs, world)
    rec = m.pack_records(mine, run(m, mine), pad_to=rows)
    assert rec.shape == (rows, shard.record_width(K)) and rec.is_cuda

    # ---- (1) torch.distributed gather over RCCL
    got = shard.gather_records(rec, force=True)
    assert (got is None) == (rank != 0)

    # ---- (2) the C-ABI collective on its own communicator
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["/opt/rocm/lib/librccl.so"]
    rccl = next((ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL) for c in cands if os.path.exists(c)), None)
    assert rccl is not None, "no RCCL library found"

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid, comm = UniqueId(), ctypes.c_void_p()
    if rank == 0:
        assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    raw = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(dev)
    dist.broadcast(raw, src=0)
    ctypes.memmove(ctypes.byref(uid), bytes(raw.cpu().numpy().tobytes()), 128)
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) == 0
    try:
        eng = m._shared.get_engine([0, 1])
        got_c = torch.full((world * rows, rec.shape[1]), -7, dtype=torch.int32, device=dev) if rank == 0 else None
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = eng.lib.imx_gather_records(eng.handle, ctypes.c_void_p(rec.data_ptr()), rec.shape[0], rec.shape[1],
                                        ctypes.c_void_p(got_c.data_ptr()) if rank == 0 else None, 0, comm, st)
        assert rc == 0, eng.lib.imx_last_error(eng.handle).decode()
        torch.cuda.synchronize()
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)

    ok = True
    if rank == 0:
        assert torch.equal(got, got_c), "imx_gather_records and the torch.distributed gather disagree"
        back = shard.sort_by_pair_id(got)
        assert shard.pair_ids_of(back).tolist() == list(range(n_pairs)), shard.pair_ids_of(back).tolist()
        # the same pairs as ONE batch on this GPU: the gathered records must be these bytes
        one = m.pack_records(list(range(n_pairs)), run(m, list(range(n_pairs))))
        diff = int((back != one).sum())
        ok = diff == 0
        print(f"MULTI-GPU {'OK' if ok else 'MISMATCH'} world={world} pairs={n_pairs} rows_per_rank={rows} backend={dist.get_backend()} "
              f"devices={torch.cuda.device_count()} differing_words={diff}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

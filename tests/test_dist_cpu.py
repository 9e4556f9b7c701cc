"""The N>1 path on CPU: two processes, gloo backend, pair sharding + record gather
(the same code path RCCL runs on the GPU box)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from image_matching_amd import shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_pairs, K, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard.shard_indices(n_pairs, rank, world)
        B = len(mine)
        g = torch.Generator().manual_seed(100 + rank)
        out = {"keypoints0": torch.rand(B, K, 2, generator=g), "keypoints1": torch.rand(B, K, 2, generator=g),
               "counts0": torch.full((B,), K, dtype=torch.int32), "counts1": torch.full((B,), K, dtype=torch.int32),
               "matches0": torch.randint(-1, K, (B, K), generator=g), "matches1": torch.randint(-1, K, (B, K), generator=g),
               "matching_scores0": torch.rand(B, K, generator=g), "matching_scores1": torch.rand(B, K, generator=g)}
        rec = shard.pack_records(mine, out)
        allrec = shard.sort_by_pair_id(shard.gather_records(rec))
        back = shard.unpack_records(allrec)
        ok = back["pair_id"].tolist() == list(range(n_pairs))
        # my own records must come back unchanged at my pair ids
        sel = torch.tensor(mine)
        ok = ok and torch.equal(back["matches0"][sel], out["matches0"]) and torch.equal(back["keypoints1"][sel], out["keypoints1"])
        q.put((rank, ok, allrec.shape))
    finally:
        dist.destroy_process_group()


def test_two_rank_pair_sharding_and_gather():
    world, n_pairs, K = 2, 8, 32
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert all(tuple(r[2]) == (n_pairs, shard.record_width(K)) for r in res)

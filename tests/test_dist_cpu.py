"""The N>1 path on CPU: several processes, gloo backend, pair sharding + the one record gather
(the same code path RCCL runs on the GPU box): 2 ranks with ragged shards, and the exact C4 shape
(BASELINE.json configs[3]: 512 pairs over 8 ranks -> 64 pairs per rank)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from image_matching_amd import shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_out(B, K, rank):
    g = torch.Generator().manual_seed(100 + rank)
    big = 1 << 24                # indices past 2^24 are not representable in float32 VALUES: the records carry bit patterns
    return {"keypoints0": torch.rand(B, K, 2, generator=g), "keypoints1": torch.rand(B, K, 2, generator=g),
            "counts0": torch.full((B,), K, dtype=torch.int32), "counts1": torch.full((B,), K, dtype=torch.int32),
            "matches0": torch.randint(-1, K, (B, K), generator=g) + (torch.arange(K) == 0) * (big + 1),
            "matches1": torch.randint(-1, K, (B, K), generator=g),
            "matching_scores0": torch.rand(B, K, generator=g), "matching_scores1": torch.rand(B, K, generator=g)}


def _worker(rank, world, port, n_pairs, K, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard.shard_indices(n_pairs, rank, world)
        B = len(mine)
        out = _fake_out(B, K, rank)
        if mode == "unpadded":          # unequal shards without padding must be refused BEFORE the collective
            try:
                shard.gather_records(shard.pack_records(mine, out))
                q.put((rank, False, "no error"))
            except ValueError as e:
                q.put((rank, "pad every shard" in str(e), str(e)))
            return
        rows = shard.shard_rows(n_pairs, world)
        rec = shard.pack_records(mine, out, pad_to=rows)
        assert rec.shape == (rows, shard.record_width(K))
        got = shard.gather_records(rec, all_ranks=(mode == "all"))
        if got is None:                 # gather to rank 0: the other ranks hold nothing
            q.put((rank, rank != 0 and mode == "gather", None))
            return
        assert got.shape == (world * rows, shard.record_width(K))
        back = shard.unpack_records(shard.sort_by_pair_id(got))
        ok = back["pair_id"].tolist() == list(range(n_pairs))            # every pair once, padding dropped
        sel = torch.tensor(mine)
        ok = ok and torch.equal(back["matches0"][sel], out["matches0"]) and torch.equal(back["keypoints1"][sel], out["keypoints1"])
        ok = ok and torch.equal(back["matching_scores1"][sel], out["matching_scores1"]) and back["matches0"].dtype == torch.int64
        ok = ok and bool((back["counts0"] == K).all())
        q.put((rank, ok, tuple(got.shape)))
    finally:
        dist.destroy_process_group()


def _run(world, n_pairs, K, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, K, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    return res


@pytest.mark.parametrize("mode", ["gather", "all"])
def test_two_rank_pair_sharding_and_gather(mode):
    """7 pairs over 2 ranks: shards of 4 and 3 -> padded to 4 rows each, padding dropped after the collective."""
    _run(2, 7, 32, mode)


def test_unequal_unpadded_shards_are_refused_with_a_clear_error():
    _run(2, 7, 8, "unpadded")


def test_c4_shape_eight_ranks_512_pairs():
    """BASELINE configs[3]: 512 pairs pair-sharded over 8 ranks = 64 per rank, one gather of the records to rank 0."""
    world, n_pairs, K = 8, 512, 16
    assert [len(shard.shard_indices(n_pairs, r, world)) for r in range(world)] == [64] * 8
    assert sorted(i for r in range(world) for i in shard.shard_indices(n_pairs, r, world)) == list(range(n_pairs))
    res = _run(world, n_pairs, K, "gather")
    assert [r[2] for r in res if r[0] == 0] == [(512, shard.record_width(K))]

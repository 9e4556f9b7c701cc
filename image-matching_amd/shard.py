"""Pair sharding across the GPUs of one node and the single collective of the path: a gather of
fixed-size match records.  Pairs are independent (the reference is batch-1 per
pair, superpoint_glue_test.py:66,72-78), so pair i goes to rank i mod world and no data-path
collective is needed until the results are collected; on MI355X the gather is one RCCL
gather over xGMI to the rank that writes the results (`backend="nccl"` is RCCL on ROCm).
world_size == 1 needs no process group.

Record = one row of 32-bit words per pair (`record_width(K)` words):
  [pair_id i32 | n0 i32 | n1 i32 | kpts0 2K f32 | kpts1 2K f32 | matches0 K i32 | matches1 K i32 |
   mscores0 K f32 | mscores1 K f32]
The buffer's torch dtype is int32; the float fields are float32 *bit patterns* viewed as int32
(`Tensor.view(dtype)`), never value-converted, so every field round-trips exactly (an int32 -1
viewed as float32 would be a NaN pattern: buffers would stop comparing equal to themselves).  A row
with pair_id == -1 is padding (ranks with fewer pairs than ceil(n_pairs / world)) and is dropped
after the gather.
"""
import torch
import torch.distributed as dist

PAD_ID = -1


def shard_indices(n_pairs, rank, world):
    """Indices of the pairs rank `rank` processes (round-robin: pair i -> rank i % world)."""
    return list(range(rank, n_pairs, world))


def shard_rows(n_pairs, world):
    """Rows every rank contributes to the gather: ceil(n_pairs / world) (short shards are padded)."""
    return (n_pairs + world - 1) // world


def record_width(K):
    """32-bit words per pair record (layout in the module docstring)."""
    return 3 + 8 * K


def _f32_as_i32(t):
    return t.to(torch.float32).contiguous().view(torch.int32)


def pack_records(pair_ids, out, pad_to=None):
    """out: dict from Engine.match_pairs (padded (B,K,...) tensors) -> (rows, record_width) int32 buffer,
    rows = max(B, pad_to); rows past B are padding (pair_id -1, everything else 0)."""
    B, K = out["matches0"].shape
    dev = out["matches0"].device
    ids = torch.as_tensor(list(pair_ids), dtype=torch.int32, device=dev).reshape(B, 1)
    i32 = lambda t: t.to(torch.int32)
    parts = [ids, i32(out["counts0"]).reshape(B, 1), i32(out["counts1"]).reshape(B, 1),
             _f32_as_i32(out["keypoints0"].reshape(B, 2 * K)), _f32_as_i32(out["keypoints1"].reshape(B, 2 * K)),
             i32(out["matches0"]), i32(out["matches1"]),
             _f32_as_i32(out["matching_scores0"]), _f32_as_i32(out["matching_scores1"])]
    rec = torch.cat(parts, dim=1).contiguous()
    if pad_to is not None and pad_to > B:
        pad = torch.zeros(pad_to - B, rec.shape[1], dtype=rec.dtype, device=dev)
        pad[:, 0] = PAD_ID
        rec = torch.cat([rec, pad])
    return rec


def pair_ids_of(rec):
    """int64 pair ids of a record buffer."""
    return rec[:, 0].long()


def drop_padding(rec):
    return rec[pair_ids_of(rec) != PAD_ID]


def unpack_records(rec):
    """(R, record_width) record buffer -> dict of per-pair tensors (inverse of pack_records; padding dropped)."""
    rec = drop_padding(rec)
    K = (rec.shape[1] - 3) // 8

    def f32(x):
        return x.contiguous().view(torch.float32)
    o = 3
    out = {"pair_id": rec[:, 0].long(), "counts0": rec[:, 1].contiguous(), "counts1": rec[:, 2].contiguous()}
    for name, w in (("keypoints0", 2 * K), ("keypoints1", 2 * K), ("matches0", K), ("matches1", K),
                    ("matching_scores0", K), ("matching_scores1", K)):
        out[name] = rec[:, o:o + w]
        o += w
    out["keypoints0"] = f32(out["keypoints0"]).reshape(-1, K, 2)
    out["keypoints1"] = f32(out["keypoints1"]).reshape(-1, K, 2)
    out["matching_scores0"], out["matching_scores1"] = f32(out["matching_scores0"]), f32(out["matching_scores1"])
    out["matches0"] = out["matches0"].long()
    out["matches1"] = out["matches1"].long()
    return out


def gather_records(rec, group=None, force=False, dst=0, all_ranks=False, check=True):
    """The one collective of the path.  Every rank contributes a (rows, w) record buffer with the SAME `rows`
    (pack_records(..., pad_to=shard_rows(n_pairs, world)); checked with a clear error before the collective).
    Default: a gather to rank `dst` -- returns the (world*rows, w) buffer ordered by rank there and None on the
    other ranks (north_star: "RCCL gather of match results").  all_ranks=True: all_gather, every rank gets it.
    Without an initialised process group (world 1) returns `rec`; `force` runs the collective even at world 1
    (bring-up check of the RCCL path on a 1-GPU box).  check=False skips the row-count verification (one tiny
    all-reduce + a host sync) for callers whose shards are equal by construction (bench.py: world * B pairs)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return rec
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if dist.get_backend(group) != "nccl" and rec.is_cuda:      # host-side gather (gloo bring-up); RCCL gathers in HBM
        out = gather_records(rec.cpu(), group, force, dst, all_ranks, check)
        return out.to(rec.device) if out is not None else None
    # equal row counts are a precondition of the fixed-size collective: verify instead of hanging / erroring inside RCCL
    if check:
        rows = torch.tensor([rec.shape[0], -rec.shape[0]], dtype=torch.int64, device=rec.device)
        dist.all_reduce(rows, op=dist.ReduceOp.MAX, group=group)
        if int(rows[0]) != -int(rows[1]):
            raise ValueError(f"gather_records: ranks contribute between {-int(rows[1])} and {int(rows[0])} rows; pad "
                             f"every shard with pack_records(..., pad_to=shard_rows(n_pairs, world))")
    rec = rec.contiguous()
    if all_ranks:
        out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(out, rec, group=group)
        return out
    dst_global = dist.get_global_rank(group, dst) if group is not None else dst
    if rank == dst:
        out = torch.empty((world, rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
        dist.gather(rec, list(out.unbind(0)), dst=dst_global, group=group)
        return out.reshape(world * rec.shape[0], rec.shape[1])
    dist.gather(rec, None, dst=dst_global, group=group)
    return None


def sort_by_pair_id(rec):
    rec = drop_padding(rec)
    return rec[torch.argsort(pair_ids_of(rec))]

"""Pair sharding across the GPUs of one node and the single collective of the path: a gather of
fixed-size match records.  Pairs are independent (the reference is batch-1 per
pair, superpoint_glue_test.py:66,72-78), so pair i goes to rank i mod world and no data-path
collective is needed until the results are collected; on MI355X the gather is one RCCL
all_gather over xGMI (`backend="nccl"` is RCCL on ROCm).  world_size == 1 needs no process group.
"""
import torch
import torch.distributed as dist


def shard_indices(n_pairs, rank, world):
    """Indices of the pairs rank `rank` processes (round-robin: pair i -> rank i % world)."""
    return list(range(rank, n_pairs, world))


def record_width(K):
    """float32 words per pair record: pair_id, n0, n1, kpts0 (2K), kpts1 (2K), matches0 (K),
    matches1 (K), mscores0 (K), mscores1 (K).  Match indices < 2^24 are exact in float32."""
    return 3 + 8 * K


def pack_records(pair_ids, out):
    """out: dict from Engine.match_pairs (padded (B,K,...) tensors) -> (B, record_width) float32."""
    B, K = out["matches0"].shape
    dev = out["matches0"].device
    ids = torch.as_tensor(pair_ids, dtype=torch.float32, device=dev).reshape(B, 1)
    parts = [ids, out["counts0"].reshape(B, 1).float(), out["counts1"].reshape(B, 1).float(),
             out["keypoints0"].reshape(B, 2 * K), out["keypoints1"].reshape(B, 2 * K),
             out["matches0"].float(), out["matches1"].float(),
             out["matching_scores0"], out["matching_scores1"]]
    return torch.cat(parts, dim=1).contiguous()


def unpack_records(rec):
    """(R, record_width) float32 -> dict of per-pair tensors (inverse of pack_records)."""
    K = (rec.shape[1] - 3) // 8
    o = 3
    out = {"pair_id": rec[:, 0].long(), "counts0": rec[:, 1].int(), "counts1": rec[:, 2].int()}
    for name, w in (("keypoints0", 2 * K), ("keypoints1", 2 * K), ("matches0", K), ("matches1", K),
                    ("matching_scores0", K), ("matching_scores1", K)):
        out[name] = rec[:, o:o + w]
        o += w
    out["keypoints0"] = out["keypoints0"].reshape(-1, K, 2)
    out["keypoints1"] = out["keypoints1"].reshape(-1, K, 2)
    out["matches0"] = out["matches0"].long()
    out["matches1"] = out["matches1"].long()
    return out


def gather_records(rec, group=None, force=False):
    """All ranks contribute (B, w) records (same B on every rank); returns (world*B, w) on every
    rank, ordered by rank.  Without an initialised process group (world 1) returns `rec`; `force`
    runs the collective even at world 1 (bring-up check of the RCCL path on a 1-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return rec
    world = dist.get_world_size(group)
    if dist.get_backend(group) != "nccl" and rec.is_cuda:      # host-side gather (gloo bring-up); RCCL gathers in HBM
        return gather_records(rec.cpu(), group).to(rec.device)
    out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec.contiguous(), group=group)
    return out


def sort_by_pair_id(rec):
    return rec[torch.argsort(rec[:, 0])]

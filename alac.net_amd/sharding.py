"""Packet-batch sharding across the GPUs of one node.

Every ALAC packet decodes independently (the bit cursor is reset per DecodeFrame call,
AlacFile.cs:432-434; coefficients are re-read per frame), so a batch shards as contiguous packet
ranges with no exchange during decode.  The only collective is the optional all-gather of decoded
PCM (north_star): equal-size shards of fixed-stride slots -> one `all_gather_into_tensor` (RCCL over
xGMI when the backend is nccl; gloo on CPU for the tests).
"""
import numpy as np


def shard_range(n_packets, rank, world):
    """Contiguous range [lo, hi) of rank `rank`; ranges differ by at most one packet."""
    lo = (n_packets * rank) // world
    hi = (n_packets * (rank + 1)) // world
    return lo, hi


def shard_batch(blob, offsets, sizes, cfg_idx, rank, world):
    """Slice a host batch down to this rank's packets (blob is shared, offsets stay absolute)."""
    lo, hi = shard_range(len(sizes), rank, world)
    ci = None if cfg_idx is None else np.ascontiguousarray(cfg_idx[lo:hi])
    return blob, np.ascontiguousarray(offsets[lo:hi]), np.ascontiguousarray(sizes[lo:hi]), ci, (lo, hi)


def padded_shard(n_packets, world):
    """Packets per rank after padding to equal shards (all_gather needs equal counts)."""
    return (n_packets + world - 1) // world


def allgather_pcm(local_pcm, n_packets, group=None):
    """All-gather equal-size PCM shards (torch tensors [per_rank, slot]) and trim the padding.
    Works with any torch.distributed backend (nccl = RCCL on ROCm, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per = local_pcm.shape[0]
    out = torch.empty((world * per,) + tuple(local_pcm.shape[1:]), dtype=local_pcm.dtype, device=local_pcm.device)
    dist.all_gather_into_tensor(out, local_pcm.contiguous(), group=group)
    # undo the padding: rank r owns packets shard_range(n_packets, r, world)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_packets, r, world)
        parts.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(parts, dim=0)

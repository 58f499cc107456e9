"""Packet-batch sharding across the GPUs of one node.

Every ALAC packet decodes independently (the bit cursor is reset per DecodeFrame call,
AlacFile.cs:432-434; coefficients are re-read per frame), so a batch shards as contiguous packet
ranges with no exchange during decode.  The only collective is the optional all-gather of decoded
PCM (north_star): equal-size shards of fixed-stride slots -> one `all_gather_into_tensor` (RCCL over
xGMI when the backend is nccl; gloo on CPU for the tests).
"""
import numpy as np


def shard_ranges(sizes, world):
    """first[world + 1]: rank r owns packets first[r] .. first[r+1].  Contiguous ranges, cut at multiples of 8 packets (the
    kernels work in groups of 8): equal packet COUNTS where that leaves the ranges' bytes within 5 % of each other (equal
    shards gather with one plain all-gather), else balanced by cumulative packet BYTES ("if packet
    sizes are skewed").  Same arithmetic as the library's alacgpu_shard_ranges (tests/test_distributed_gloo.py compares
    the two); kept in Python too so that a rank can partition before it loads anything."""
    sizes = np.asarray(sizes, dtype=np.uint64)
    n = len(sizes)
    first = np.array([min(n, ((n * r // world) + 7) & ~7) for r in range(world + 1)], dtype=np.uint32)
    first[0], first[world] = 0, n
    by = [int(sizes[first[r]:first[r + 1]].sum()) for r in range(world)]
    if max(by) * 100 <= min(by) * 105:
        return first
    total = int(sizes.sum())
    groups = [int(sizes[i:i + 8].sum()) for i in range(0, n, 8)]
    acc, gi = 0, 0
    for r in range(1, world):
        want = total * r // world
        while gi < len(groups) and acc + groups[gi] // 2 <= want:
            acc += groups[gi]
            gi += 1
        first[r] = min(n, gi * 8)
    first[world] = n
    return np.maximum.accumulate(first)


def shard_range(n_packets, rank, world, sizes=None):
    """Contiguous range [lo, hi) of rank `rank`: by packet bytes when `sizes` is given (shard_ranges), else by count."""
    if sizes is not None:
        first = shard_ranges(sizes, world)
        return int(first[rank]), int(first[rank + 1])
    lo = (n_packets * rank) // world
    hi = (n_packets * (rank + 1)) // world
    return lo, hi


def shard_batch(blob, offsets, sizes, cfg_idx, rank, world, by_bytes=False):
    """Slice a host batch down to this rank's packets (blob is shared, offsets stay absolute)."""
    lo, hi = shard_range(len(sizes), rank, world, sizes if by_bytes else None)
    ci = None if cfg_idx is None else np.ascontiguousarray(cfg_idx[lo:hi])
    return blob, np.ascontiguousarray(offsets[lo:hi]), np.ascontiguousarray(sizes[lo:hi]), ci, (lo, hi)


def padded_shard(n_packets, world, first=None):
    """Packets per rank after padding to equal shards (all_gather needs equal counts): the largest range."""
    if first is not None:
        return int(max(int(first[r + 1]) - int(first[r]) for r in range(world)))
    return (n_packets + world - 1) // world


def allgather_pcm(local_pcm, n_packets, group=None, first=None):
    """All-gather equal-size (padded) PCM shards (torch tensors [per_rank, slot]) and trim the padding.  `first`: the
    partition (shard_ranges) when it is not the by-count one.  Works with any torch.distributed backend (nccl = RCCL on
    ROCm, gloo on CPU); the library's own alacgpu_allgather_pcm gathers in place without padding."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per = local_pcm.shape[0]
    out = torch.empty((world * per,) + tuple(local_pcm.shape[1:]), dtype=local_pcm.dtype, device=local_pcm.device)
    dist.all_gather_into_tensor(out, local_pcm.contiguous(), group=group)
    # undo the padding: rank r owns packets first[r] .. first[r+1] (by count: shard_range(n_packets, r, world))
    parts = []
    for r in range(world):
        lo, hi = (int(first[r]), int(first[r + 1])) if first is not None else shard_range(n_packets, r, world)
        parts.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(parts, dim=0)


class ChunkedDecodeAllGather:
    """Decode + all-gather of one rank's shard, overlapped: the shard is decoded in `n_chunks` contiguous packet ranges
    and the all-gather of range k runs (on the collective's own stream) while range k+1 decodes.

    `local_pcm`: this rank's [per_rank, slot] PCM tensor on the GPU (the decode writes into it).  With a CUDA-aware
    backend (nccl = RCCL over xGMI) the operands stay in HBM; otherwise (gloo rehearsal) every range goes through the
    host.  `run(decode_range)` calls decode_range(lo, hi) for every range -- which must enqueue the decode of packets
    [lo, hi) of the shard on torch's current stream -- and returns the gathered [world * per_rank, slot] tensor in
    global packet order (rank r's packets at [r * per_rank, (r + 1) * per_rank)).

    RCCL picks the all-gather algorithm itself; with equal-size shards of 32 MiB and more per range it is bandwidth
    bound on the 7 xGMI links of every GPU.  No collective touches the decode itself."""

    def __init__(self, local_pcm, world, n_chunks=4, cuda_collective=True, group=None):
        import torch

        self.local, self.world, self.group, self.cuda = local_pcm, world, group, cuda_collective
        per = local_pcm.shape[0]
        n_chunks = max(1, min(n_chunks, per))
        self.bounds = [(per * k // n_chunks, per * (k + 1) // n_chunks) for k in range(n_chunks)]
        cdev = local_pcm.device if cuda_collective else torch.device("cpu")
        tail = tuple(local_pcm.shape[1:])
        self.full = torch.empty((world * per,) + tail, dtype=local_pcm.dtype, device=cdev)
        self.stage = [torch.empty((world * (hi - lo),) + tail, dtype=local_pcm.dtype, device=cdev) for lo, hi in self.bounds]

    def run(self, decode_range):
        import torch.distributed as dist

        per = self.local.shape[0]
        works = []
        for k, (lo, hi) in enumerate(self.bounds):
            decode_range(lo, hi)
            src = self.local[lo:hi] if self.cuda else self.local[lo:hi].cpu()
            # the collective starts once the current stream has reached this point, i.e. after range k's decode; the
            # current stream itself goes on with range k+1
            works.append(dist.all_gather_into_tensor(self.stage[k], src.contiguous(), group=self.group, async_op=True))
        fv = self.full.view((self.world, per) + tuple(self.full.shape[1:]))
        for k, (lo, hi) in enumerate(self.bounds):
            works[k].wait()
            fv[:, lo:hi].copy_(self.stage[k].view((self.world, hi - lo) + tuple(self.full.shape[1:])))
        return self.full

"""N>1 path on CPU: world_size-2 gloo processes shard a batch by packet range, decode their shard
(the CPU oracle stands in for the GPU decode here: this test is about the partitioning + the PCM
all-gather, which is backend-agnostic), all-gather the PCM and compare with the single-rank result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_packets, outdir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alac.net_amd import sharding, synth
    import alac_oracle_py as orc

    b = synth.make_config_batch(5, n_packets=n_packets, n_threads=1)  # identical on every rank (seeded)
    blob, offs, sizes, ci, (lo, hi) = sharding.shard_batch(b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], rank, world)
    pcm, ob, os_, st = orc.decode_batch(orc.make_cfgs(b["stream_cfgs"]), blob, offs, sizes, ci, b["slot_ints"])
    per = sharding.padded_shard(n_packets, world)
    local = torch.zeros((per, b["slot_ints"]), dtype=torch.int32)
    local[: hi - lo] = torch.from_numpy(pcm)
    full = sharding.allgather_pcm(local, n_packets)
    assert full.shape == (n_packets, b["slot_ints"])
    # the chunk-overlapped pipeline bench.py uses at N>1 (decode range by range, gather each range behind it)
    local2 = torch.zeros_like(local)

    def decode_range(l, h):          # stand-in decode of packets [l, h) of this rank's shard
        h2 = min(h, hi - lo)
        if l < h2:
            local2[l:h2] = torch.from_numpy(pcm[l:h2])

    pipe = sharding.ChunkedDecodeAllGather(local2, world, n_chunks=3, cuda_collective=False)
    full2 = pipe.run(decode_range)
    parts = [full2[r * per: r * per + (sharding.shard_range(n_packets, r, world)[1] - sharding.shard_range(n_packets, r, world)[0])]
             for r in range(world)]
    assert torch.equal(torch.cat(parts, dim=0), full)
    # the byte-balanced partition (what the library's multi-GPU entry points use): skew the sizes so that it differs from the
    # by-count one, decode those ranges, gather with the same partition
    skew = b["sizes"].astype(np.uint64) * np.where(np.arange(n_packets) < n_packets // 2, 1, 7)
    first = sharding.shard_ranges(skew, world)
    lo3, hi3 = int(first[rank]), int(first[rank + 1])
    ci3 = None if b["cfg_idx"] is None else np.ascontiguousarray(b["cfg_idx"][lo3:hi3])
    pcm3 = orc.decode_batch(orc.make_cfgs(b["stream_cfgs"]), b["blob"], np.ascontiguousarray(b["offsets"][lo3:hi3]),
                            np.ascontiguousarray(b["sizes"][lo3:hi3]), ci3, b["slot_ints"])[0] if hi3 > lo3 else np.zeros((0, b["slot_ints"]), np.int32)
    per3 = sharding.padded_shard(n_packets, world, first)
    local3 = torch.zeros((per3, b["slot_ints"]), dtype=torch.int32)
    local3[: hi3 - lo3] = torch.from_numpy(pcm3)
    full3 = sharding.allgather_pcm(local3, n_packets, first=first)
    assert torch.equal(full3, full)
    if rank == 0:
        ref = orc.decode_batch(orc.make_cfgs(b["stream_cfgs"]), b["blob"], b["offsets"], b["sizes"], b["cfg_idx"],
                               b["slot_ints"])[0]
        ok = bool(np.array_equal(full.numpy(), ref))
        open(os.path.join(outdir, "result.txt"), "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_packets", [10, 7])  # 7: uneven shards, exercises the padding
def test_two_rank_shard_and_allgather(tmp_path, n_packets):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_packets, str(tmp_path)), nprocs=world, join=True)
    assert open(tmp_path / "result.txt").read() == "ok"


def test_byte_balanced_ranges_python_and_library_agree():
    """ranges balanced by cumulative packet bytes when packet sizes are skewed; the library's
    alacgpu_shard_ranges (what alacgpu_decode_batch_sharded and the RCCL entry points use) and sharding.shard_ranges (what the
    Python ranks use) are the same arithmetic."""
    import alac.net_amd as pkg
    from alac.net_amd import sharding, synth

    b = synth.make_config_batch(5, n_packets=4096, n_threads=2)
    rng = np.random.default_rng(11)
    skew = b["sizes"].copy()
    skew[:1500] = 3                                    # a long run of tiny packets in front
    cases = [b["sizes"], skew, rng.integers(1, 70000, 999).astype(np.uint32), np.array([5, 5, 5], dtype=np.uint32),
             np.zeros(0, dtype=np.uint32), np.full(65536, 5000, dtype=np.uint32)]
    for sizes in cases:
        for world in (1, 2, 3, 4, 8):
            a = pkg.shard_ranges(sizes, world)
            c = sharding.shard_ranges(sizes, world)
            assert np.array_equal(a, c), (len(sizes), world, a, c)
            assert a[0] == 0 and a[-1] == len(sizes) and np.all(np.diff(a.astype(np.int64)) >= 0)
            assert all(int(x) % 8 == 0 or int(x) == len(sizes) for x in a[1:-1])   # whole groups of 8
            if len(sizes) >= 4096:
                by = [int(sizes[a[r]:a[r + 1]].astype(np.int64).sum()) for r in range(world)]
                assert max(by) <= 1.05 * min(by), (world, by)
    # uniform packets: equal counts (one plain all-gather)
    a = pkg.shard_ranges(np.full(65536, 5000, dtype=np.uint32), 8)
    assert np.array_equal(a, np.arange(9) * 8192)
    # shard_batch by bytes hands out exactly those ranges
    for rank in range(8):
        _, offs, sz, ci, (lo, hi) = sharding.shard_batch(b["blob"], b["offsets"], skew, b["cfg_idx"], rank, 8, by_bytes=True)
        f = sharding.shard_ranges(skew, 8)
        assert (lo, hi) == (int(f[rank]), int(f[rank + 1])) and len(sz) == hi - lo


def test_sharded_entry_rejects_the_same_context_twice():
    """alacgpu_decode_batch_sharded: one context on two threads would race (a context is not thread-safe): checked before
    anything touches a GPU -- needs no device, the handles are never dereferenced past the comparison."""
    import ctypes as C
    import alac.net_amd as pkg

    L = pkg.lib()
    fake = C.c_void_p(0x1000)
    handles = (C.c_void_p * 2)(fake, fake)
    # (n_ctxs == 2 with identical handles: BAD_ARG comes from the duplicate check or, without one, a crash)
    dup = L.alacgpu_decode_batch_sharded(handles, 2, None, 0, None, None, None, 0, None, 0, None, None, None)
    assert dup == -1


def test_shard_ranges_cover_everything():
    from alac.net_amd import sharding

    for n in (0, 1, 7, 4096, 65536):
        for w in (1, 2, 4, 8):
            r = [sharding.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1

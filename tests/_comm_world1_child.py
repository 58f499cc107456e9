"""Child of tests/test_comm.py (GPU box): RCCL with world size 1 -- the native entry points (alacgpu_comm_*) and the
torch.distributed nccl path of alac.net_amd/sharding.py -- on one GPU.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import torch
import torch.distributed as dist

import alac.net_amd as pkg
from alac.net_amd import sharding, synth
import alac_oracle_py as orc   # the checker

port = sys.argv[1]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
out = {}
b = synth.make_config_batch(5, n_packets=256, want_pcm=False)
n, slot, nb = 256, int(b["slot_ints"]), int(b["blob"].size)
ref = orc.decode_batch(orc.make_cfgs(b["stream_cfgs"]), b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], slot)
d_blob = torch.zeros((nb + 63) // 16 * 16 + 64, dtype=torch.uint8, device=dev)
d_blob[:nb] = torch.from_numpy(b["blob"]).to(dev)
d_off = torch.from_numpy(b["offsets"].astype(np.int64)).to(dev)
d_sz = torch.from_numpy(b["sizes"].astype(np.int32)).to(dev)
d_ci = torch.from_numpy(b["cfg_idx"].astype(np.int16)).to(dev)


def same(pcm, st):
    pcm, st = pcm.cpu().numpy(), st.cpu().numpy()
    if not np.array_equal(st, ref[3]):
        return False
    for p in range(n):
        cnt = int(ref[2][p]) * int(b["stream_cfgs"][int(b["cfg_idx"][p])][5])
        if ref[3][p] == 0 and not np.array_equal(pcm[p, :cnt], ref[0][p, :cnt]):
            return False
    return True


with pkg.AlacGpuContext(b["stream_cfgs"], device=0) as ctx:
    # ---- native: RCCL through the C ABI, one rank ----
    uid = pkg.AlacGpuComm.unique_id()
    out["uid_nonzero"] = bool(np.any(uid != 0))
    first = pkg.shard_ranges(b["sizes"], 1)
    with pkg.AlacGpuComm(ctx, uid, 0, 1) as comm:
        side = torch.cuda.Stream(dev)            # a stream of the caller's own: ordering must hold there
        full = torch.zeros((n, slot), dtype=torch.int32, device=dev)
        ob = torch.zeros(n, dtype=torch.int32, device=dev); os_ = torch.zeros_like(ob); st = torch.full_like(ob, -1)
        torch.cuda.synchronize(dev)
        ctx.decode_batch_device(d_blob, nb, d_off, d_sz, d_ci, n, full, slot, ob, os_, st, stream=side.cuda_stream)
        comm.allgather_pcm(full, first, slot, stream=side.cuda_stream)       # degenerate gather, in place, behind the decode
        side.synchronize()
        out["native_allgather_world1"] = same(full, st)
        for chunks in (1, 3, 4):
            full.zero_(); st.fill_(-1)
            torch.cuda.synchronize(dev)
            comm.decode_allgather_device(d_blob, nb, d_off, d_sz, d_ci, first, full, slot, ob, os_, st, n_chunks=chunks,
                                         stream=side.cuda_stream)
            # work enqueued on the SAME stream behind the call must see the gathered PCM (the call makes the caller's stream
            # wait for the collective's stream)
            with torch.cuda.stream(side):
                copy = full.clone()
            side.synchronize()
            out[f"native_decode_allgather_chunks{chunks}"] = same(copy, st)
        out["comm_rank_world"] = [pkg.lib().alacgpu_comm_rank(comm._comm), pkg.lib().alacgpu_comm_world(comm._comm)]
    # ---- torch.distributed, backend nccl (= RCCL), one rank: the path bench.py takes when the native one is unavailable ----
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", device_id=dev)
    pcm = torch.zeros((n, slot), dtype=torch.int32, device=dev)
    ob = torch.zeros(n, dtype=torch.int32, device=dev); os_ = torch.zeros_like(ob); st = torch.full_like(ob, -1)
    cur = torch.cuda.current_stream(dev)

    def decode_range(lo, hi):
        ctx.decode_batch_device(d_blob, nb, d_off[lo:hi], d_sz[lo:hi], d_ci[lo:hi], hi - lo, pcm[lo:hi], slot, ob[lo:hi], os_[lo:hi],
                                st[lo:hi], stream=cur.cuda_stream)
    decode_range(0, n)
    g = sharding.allgather_pcm(pcm, n)
    torch.cuda.synchronize(dev)
    out["torch_nccl_allgather_world1"] = same(g, st)
    pcm.zero_(); st.fill_(-1)
    pipe = sharding.ChunkedDecodeAllGather(pcm, 1, n_chunks=4, cuda_collective=True)
    full = pipe.run(decode_range)
    torch.cuda.synchronize(dev)
    out["torch_nccl_chunked_world1"] = same(full, st)
    dist.barrier()
    dist.destroy_process_group()
print(json.dumps(out), flush=True)

#!/usr/bin/env python3
"""Experiment (GPU box): steady-state throughput when consecutive batches go to TWO streams (the tail of one launch, where only
its slowest workgroups are left, overlaps the start of the next) against one stream.  cfg2, device-resident inputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import alac.net_amd as pkg
from alac.net_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = 40
b = synth.make_config_batch(2, n_packets=n)
dev = torch.device("cuda", 0)
slot, nb = int(b["slot_ints"]), int(b["blob"].size)
d_blob = torch.zeros((nb + 63) // 16 * 16 + 64, dtype=torch.uint8, device=dev); d_blob[:nb] = torch.from_numpy(b["blob"]).to(dev)
d_off = torch.from_numpy(b["offsets"].astype(np.int64)).to(dev); d_sz = torch.from_numpy(b["sizes"].astype(np.int32)).to(dev)
outs = []
for k in range(4):
    outs.append((torch.zeros((n, slot), dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev),
                 torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)))
streams = [torch.cuda.Stream(dev) for _ in range(4)]
with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
    def run(nstreams):
        for w in range(4):
            k = w % nstreams
            ctx.decode_batch_device(d_blob, nb, d_off, d_sz, None, n, outs[k][0], slot, outs[k][1], outs[k][2], outs[k][3], stream=streams[k].cuda_stream)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for w in range(steps):
            k = w % nstreams
            ctx.decode_batch_device(d_blob, nb, d_off, d_sz, None, n, outs[k][0], slot, outs[k][1], outs[k][2], outs[k][3], stream=streams[k].cuda_stream)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps * 1e3
    for rep in range(2):
        t = [run(k) for k in (1, 2, 3, 4)]
        print(f"{n} packets per batch, ms per batch with 1 / 2 / 3 / 4 batches in flight (one stream each): " + " / ".join(f"{x:.4f}" for x in t)
              + f"  ({(t[0] / t[1] - 1) * 100:+.1f} % / {(t[0] / t[2] - 1) * 100:+.1f} % / {(t[0] / t[3] - 1) * 100:+.1f} % throughput)")
    assert all(int(o[3].abs().max()) == 0 for o in outs)

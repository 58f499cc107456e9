#!/usr/bin/env python3
"""Experiment (GPU box): the kernels storing their PCM straight into page-locked HOST memory (zero copy) against a device
buffer + D2H copy.  cfg4 (one channel: nothing is parked in the slot, so no read-back crosses the link)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import alac.net_amd as pkg
from alac.net_amd import synth
cfgno = int(sys.argv[1]) if len(sys.argv) > 1 else 4
npk = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
b = synth.make_config_batch(cfgno, n_packets=npk)
dev = torch.device("cuda", 0)
n, slot, nb = npk, int(b["slot_ints"]), int(b["blob"].size)
d_blob = torch.zeros((nb + 63) // 16 * 16 + 64, dtype=torch.uint8, device=dev); d_blob[:nb] = torch.from_numpy(b["blob"]).to(dev)
d_off = torch.from_numpy(b["offsets"].astype(np.int64)).to(dev); d_sz = torch.from_numpy(b["sizes"].astype(np.int32)).to(dev)
d_ci = None if b["cfg_idx"] is None else torch.from_numpy(b["cfg_idx"].astype(np.int16)).to(dev)
d_pcm = torch.zeros((n, slot), dtype=torch.int32, device=dev)
h_pcm = torch.zeros((n, slot), dtype=torch.int32).pin_memory()
h_pcm2 = torch.zeros((n, slot), dtype=torch.int32).pin_memory()
d_ob = torch.zeros(n, dtype=torch.int32, device=dev); d_os = torch.zeros_like(d_ob); d_st = torch.zeros_like(d_ob)
s = torch.cuda.current_stream()
with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
    for fmt in (0, 1):
        ctx.set_output_format(fmt)
        for name, tgt in (("device buffer", d_pcm), ("pinned host (zero copy)", h_pcm)):
            for rep in range(3):
                torch.cuda.synchronize(); t = time.perf_counter()
                ctx.decode_batch_device(d_blob, nb, d_off, d_sz, d_ci, n, tgt, slot, d_ob, d_os, d_st, stream=s.cuda_stream)
                if tgt is d_pcm:
                    h_pcm2.copy_(d_pcm, non_blocking=True)
                torch.cuda.synchronize(); dt = time.perf_counter() - t
            ok = bool((d_st.cpu().numpy() == 0).all())
            same = bool(torch.equal(h_pcm, h_pcm2)) if tgt is h_pcm else None
            print(f"cfg{cfgno} {n} packets fmt {fmt} {name:<26s} kernel {ctx.last_kernel_ms():7.3f} ms  decode + PCM in host memory {dt * 1e3:7.3f} ms  ok={ok} same={same}", flush=True)

#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (alacgpu_decode_batch: upload, decode and download overlapped range
by range) on cfg2 -- a note for DESIGN.md; bench.py's `value` is the HBM-resident rate.  Rows: ranges per batch (1 = the
old copy -> kernel -> copy), output format, pinned vs ordinary caller memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import alac.net_amd as pkg
from alac.net_amd import synth

b = synth.make_config_batch(2)
n = len(b["sizes"])
slot = int(b["slot_ints"])
samples = int((b["descs"]["n"].astype(np.int64) * 2).sum())
REPS = 8
pin_pcm = pkg.PinnedBuffer((n, slot), np.int32)
pin_blob = pkg.PinnedBuffer(b["blob"].size, np.uint8)
pin_blob.array[:] = b["blob"]
pag_pcm = np.zeros((n, slot), np.int32)
for chunks in (1, 2, 4):
    os.environ["ALACGPU_HOST_CHUNKS"] = str(chunks)
    with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
        for fmt, fname in ((0, "int32"), (1, "packed LE")):
            ctx.set_output_format(fmt)
            for blob, out, mem in ((b["blob"], pag_pcm, "pageable"), (pin_blob.array, pin_pcm.array, "pinned")):
                ctx.decode_batch(blob, b["offsets"], b["sizes"], None, slot, out=out)
                t = time.perf_counter()
                for _ in range(REPS):
                    pcm, ob, os_, st = ctx.decode_batch(blob, b["offsets"], b["sizes"], None, slot, out=out)
                dt = (time.perf_counter() - t) / REPS
                print(f"ranges {chunks}  {fname:<9s} {mem:<8s} {dt * 1e3:7.2f} ms / 4096-packet batch = {samples / dt / 1e6:7.0f} Msamples/s  ok={bool((st == 0).all())}", flush=True)

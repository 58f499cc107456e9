#!/usr/bin/env python3
"""One configuration of tools/host_path_rate.py, for rocprofv3 --kernel-trace --memory-copy-trace (timeline of the host path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import alac.net_amd as pkg
from alac.net_amd import synth
fmt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
pinned = (sys.argv[2] if len(sys.argv) > 2 else "pinned") == "pinned"
b = synth.make_config_batch(2)
n, slot = len(b["sizes"]), int(b["slot_ints"])
pin_pcm = pkg.PinnedBuffer((n, slot), np.int32); pin_blob = pkg.PinnedBuffer(b["blob"].size, np.uint8); pin_blob.array[:] = b["blob"]
pag = np.zeros((n, slot), np.int32)
with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
    ctx.set_output_format(fmt)
    blob, out = (pin_blob.array, pin_pcm.array) if pinned else (b["blob"], pag)
    for _ in range(4):
        t = time.perf_counter()
        ctx.decode_batch(blob, b["offsets"], b["sizes"], None, slot, out=out)
        print("call ms", (time.perf_counter() - t) * 1e3, flush=True)

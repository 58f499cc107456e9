#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (alacgpu_decode_batch: H2D + kernel + D2H, pageable
numpy buffers) on cfg2 -- a note for DESIGN.md; bench.py's `value` is the HBM-resident rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import alac.net_amd as pkg
from alac.net_amd import synth

b = synth.make_config_batch(2)
samples = int((b["descs"]["n"].astype(np.int64) * 2).sum())
with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
    for fmt in (0, 1):
        ctx.set_output_format(fmt)
        ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], None, b["slot_ints"])
        t = time.perf_counter()
        reps = 5
        for _ in range(reps):
            pcm, ob, os_, st = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], None, b["slot_ints"])
        dt = (time.perf_counter() - t) / reps
        print(f"output_format={fmt}: {dt*1e3:.2f} ms per 4096-packet batch incl. H2D/D2H -> {samples/dt/1e6:.0f} Msamples/s; kernel {ctx.last_kernel_ms():.3f} ms; status ok={bool((st==0).all())}")

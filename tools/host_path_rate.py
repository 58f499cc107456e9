#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (alacgpu_decode_batch: H2D + kernel + D2H, blocking) on cfg2 --
a note for DESIGN.md; bench.py's `value` is the HBM-resident rate.  Rows: fresh PCM array per call vs a reused one, for
both output formats (int32 per sample / packed little-endian PCM)."""
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


def rate(ctx, out, label):
    ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], None, slot, out=out)
    t = time.perf_counter()
    for _ in range(REPS):
        pcm, ob, os_, st = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], None, slot, out=out)
    dt = (time.perf_counter() - t) / REPS
    print(f"{label:<48s} {dt * 1e3:7.2f} ms / 4096-packet batch = {samples / dt / 1e6:7.0f} Msamples/s "
          f"(kernel {ctx.last_kernel_ms():.3f} ms; ok={bool((st == 0).all())})", flush=True)


with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
    for fmt, fname in ((0, "int32"), (1, "packed LE")):
        ctx.set_output_format(fmt)
        rate(ctx, None, f"fresh PCM array each call, {fname}")
        rate(ctx, np.zeros((n, slot), np.int32), f"reused PCM array, {fname}")

#!/usr/bin/env python3
"""Per-workgroup diagnostics of the decode kernels (runs ON THE GPU BOX): start / end clocks and wave placement, through
the diagnostic entry point alacgpu_dbg_decode_batch_device_stamps (not part of the public ABI).
usage: python tools/stamps.py [--config 2] [--packets 4096]   (ALACGPU_DENSE=0/1 picks the arrangement)"""
import argparse, collections, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import alac.net_amd as pkg
from alac.net_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--packets", type=int, default=4096)
a = ap.parse_args()
b = synth.make_config_batch(a.config, n_packets=a.packets)
dev = torch.device("cuda", 0)
n, slot, nb = a.packets, int(b["slot_ints"]), int(b["blob"].size)
d_blob = torch.zeros((nb + 63) // 16 * 16 + 64, dtype=torch.uint8, device=dev); d_blob[:nb] = torch.from_numpy(b["blob"]).to(dev)
d_off = torch.from_numpy(b["offsets"].astype(np.int64)).to(dev); d_sz = torch.from_numpy(b["sizes"].astype(np.int32)).to(dev)
d_ci = None if b["cfg_idx"] is None else torch.from_numpy(b["cfg_idx"].astype(np.int16)).to(dev)
d_pcm = torch.zeros((n, slot), dtype=torch.int32, device=dev)
d_ob = torch.zeros(n, dtype=torch.int32, device=dev); d_os = torch.zeros_like(d_ob); d_st = torch.zeros_like(d_ob)
nwg = (n + 7) // 8
d_stamps = torch.zeros(8 * nwg, dtype=torch.int64, device=dev)
L = pkg.lib()
fn = L.alacgpu_dbg_decode_batch_device_stamps
fn.restype = C.c_int
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
    for rep in range(3):
        d_stamps.zero_()
        rc = fn(ctx._ctx, vp(d_blob), C.c_uint64(nb), vp(d_off), vp(d_sz), vp(d_ci), C.c_uint32(n), vp(d_pcm), C.c_uint32(slot),
                vp(d_ob), vp(d_os), vp(d_st), None, vp(d_stamps))
        assert rc == 0
        torch.cuda.synchronize()
    ms = ctx.last_kernel_ms()
s = d_stamps.cpu().numpy().astype(np.uint64).reshape(-1, 8)
s = s[s[:, 0] != 0]
t0 = s[:, 0].min()
start = (s[:, 0] - t0).astype(np.float64); end = (s[:, 2] - t0).astype(np.float64)
dur = end - start
print(f"kernel {ms:.3f} ms; {len(s)} workgroups stamped; duration cycles: min {dur.min():.0f} median {np.median(dur):.0f} mean {dur.mean():.0f} max {dur.max():.0f}; last end {end.max():.0f}")
if "_diag" in os.path.basename(os.environ.get("ALACGPU_LIB", "")):
    # diagnostic build: unit counters of the entropy wave (both passes summed), see alac_diag.h
    hi = lambda x: (x >> np.uint64(32)).astype(np.float64)
    lo = lambda x: (x & np.uint64(0xFFFFFFFF)).astype(np.float64)
    f16 = lambda x, sh: ((x >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.float64)
    cols = {"plain_ok": hi(s[:, 1]), "fail_esc": lo(s[:, 1]), "z_units": hi(s[:, 4]), "esc_units": lo(s[:, 4]),
            "fail_run": f16(s[:, 5], 48), "redo": f16(s[:, 5], 32), "full_units": f16(s[:, 6], 48), "late_run": f16(s[:, 6], 32),
            "wide_units": hi(s[:, 7]), "fail_range": lo(s[:, 7])}
    order = np.argsort(dur)
    print("per-workgroup unit counts      mean   | fastest 3 workgroups           | slowest 5 workgroups")
    for k, v in cols.items():
        print(f"  {k:<18s} {v.mean():10.1f}   | " + " ".join(f"{v[i]:9.0f}" for i in order[:3]) + "  | " + " ".join(f"{v[i]:9.0f}" for i in order[-5:]))
    print(f"  {'duration':<18s} {dur.mean():10.0f}   | " + " ".join(f"{dur[i]:9.0f}" for i in order[:3]) + "  | " + " ".join(f"{dur[i]:9.0f}" for i in order[-5:]))
    X = np.stack([np.ones(len(dur)), cols["fail_esc"], cols["esc_units"], cols["z_units"], cols["redo"], cols["full_units"]], axis=1)
    coef, *_ = np.linalg.lstsq(X, dur, rcond=None)
    print("least squares: cycles = %.0f + %.0f per failed plain unit (escape) + %.0f per escape-tier unit + %.0f per run-aware unit + %.0f per unit redone by rice_step + %.0f per general-tier unit;  residual rms %.0f"
          % (*coef, np.sqrt(np.mean((X @ coef - dur) ** 2))))

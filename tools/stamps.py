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
    # diagnostic build: unit counters of the entropy wave (both passes summed), see SpecStats in alac_kernels.hip
    hi = lambda x: (x >> np.uint64(32)).astype(np.float64)
    lo = lambda x: (x & np.uint64(0xFFFFFFFF)).astype(np.float64)
    f16 = lambda x, sh: ((x >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.float64)
    cols = {"plain_ok": hi(s[:, 1]), "fail_esc": lo(s[:, 1]), "z_units": hi(s[:, 4]), "esc_units": lo(s[:, 4]),
            "fail_run": f16(s[:, 5], 48), "redo": f16(s[:, 5], 32), "full_units": f16(s[:, 6], 48), "late_run": f16(s[:, 6], 32),
            "fir_barrier_wait": s[:, 7].astype(np.float64),   # (first and last chunks of a pass only: the FIR wave's common chunks carry no stamps)
            # cycle accounts of the entropy wave (both passes): inside the 8 steps of plain units, at chunk barriers, pass set-up,
            # chunks decoded by the generic step; what is left of `duration` is the code between units and the stamps themselves
            "cyc_plain_steps": lo(s[:, 3]), "cyc_barrier_wait": hi(s[:, 3]), "cyc_pass_setup": lo(s[:, 6]), "cyc_generic_chunks": lo(s[:, 5])}
    order = np.argsort(dur)
    print("per-workgroup unit counts      mean   | fastest 3 workgroups           | slowest 5 workgroups")
    for k, v in cols.items():
        print(f"  {k:<18s} {v.mean():10.1f}   | " + " ".join(f"{v[i]:9.0f}" for i in order[:3]) + "  | " + " ".join(f"{v[i]:9.0f}" for i in order[-5:]))
    print(f"  {'duration':<18s} {dur.mean():10.0f}   | " + " ".join(f"{dur[i]:9.0f}" for i in order[:3]) + "  | " + " ".join(f"{dur[i]:9.0f}" for i in order[-5:]))
    X = np.stack([np.ones(len(dur)), cols["fail_esc"], cols["esc_units"], cols["z_units"], cols["redo"], cols["full_units"]], axis=1)
    coef, res, *_ = np.linalg.lstsq(X, dur, rcond=None)
    pred = X @ coef
    print("least squares: cycles = %.0f + %.0f per failed plain unit (escape) + %.0f per escape-tier unit + %.0f per run-aware unit + %.0f per "
          "unit redone by rice_step + %.0f per general-tier unit;  residual rms %.0f" % (*coef, np.sqrt(np.mean((dur - pred) ** 2))))
    sys.exit(0)
pat = s[:, 3]
if pat.any():
    cu = (pat >> np.uint64(32)).astype(np.int64)
    simds = [((pat >> np.uint64(4 * i)) & np.uint64(15)).astype(np.int64) for i in range(5)]
    dbl = []
    for i in range(len(s)):
        cnt = collections.Counter(int(x[i]) for x in simds)
        dbl.append(max(cnt, key=cnt.get) if max(cnt.values()) == 2 and len(cnt) == 4 else -1)
    dbl = np.array(dbl)
    print("doubled-SIMD histogram:", dict(collections.Counter(dbl.tolist())))
    per_cu = collections.defaultdict(list)
    for i in range(len(s)):
        per_cu[int(cu[i])].append((start[i], end[i], int(dbl[i]), [int(x[i]) for x in simds]))
    print("CUs used:", len(per_cu), " workgroups per CU:", dict(collections.Counter(len(v) for v in per_cu.values())))
    for c in list(per_cu)[:3]:
        print(" CU", c)
        for st_, en, d, sm in sorted(per_cu[c]):
            print(f"   start {st_:9.0f} end {en:9.0f} dur {en - st_:9.0f}  doubled SIMD {d}  waves on SIMDs {sm}")

#!/usr/bin/env python3
"""Experiment (GPU box): does a packet's SIZE predict how long its workgroup takes?  Per-workgroup duration (stamps of the
diagnostic entry point) against the summed / largest packet size of the workgroup's 8 packets, cfg2."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import alac.net_amd as pkg
from alac.net_amd import synth
n = 4096
b = synth.make_config_batch(2, n_packets=n)
dev = torch.device("cuda", 0)
slot, nb = int(b["slot_ints"]), int(b["blob"].size)
d_blob = torch.zeros((nb + 63) // 16 * 16 + 64, dtype=torch.uint8, device=dev); d_blob[:nb] = torch.from_numpy(b["blob"]).to(dev)
d_off = torch.from_numpy(b["offsets"].astype(np.int64)).to(dev); d_sz = torch.from_numpy(b["sizes"].astype(np.int32)).to(dev)
d_pcm = torch.zeros((n, slot), dtype=torch.int32, device=dev)
d_ob = torch.zeros(n, dtype=torch.int32, device=dev); d_os = torch.zeros_like(d_ob); d_st = torch.zeros_like(d_ob)
d_stamps = torch.zeros(8 * (n // 8), dtype=torch.int64, device=dev)
L = pkg.lib(); fn = L.alacgpu_dbg_decode_batch_device_stamps; fn.restype = C.c_int
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
    for rep in range(3):
        d_stamps.zero_()
        assert fn(ctx._ctx, vp(d_blob), C.c_uint64(nb), vp(d_off), vp(d_sz), None, C.c_uint32(n), vp(d_pcm), C.c_uint32(slot), vp(d_ob), vp(d_os), vp(d_st), None, vp(d_stamps)) == 0
        torch.cuda.synchronize()
s = d_stamps.cpu().numpy().astype(np.uint64).reshape(-1, 8)
dur = (s[:, 2] - s[:, 0]).astype(np.float64)
sz = b["sizes"].astype(np.float64).reshape(-1, 8)
print("packet size: min %.0f median %.0f max %.0f" % (sz.min(), np.median(sz), sz.max()))
for name, x in (("sum of sizes", sz.sum(1)), ("largest size", sz.max(1))):
    print(f"corr(duration, {name}) = {np.corrcoef(dur, x)[0, 1]:.3f}")
order = np.argsort(dur)
print("slowest 8 workgroups: duration, packet sizes")
for i in order[-8:]:
    print(f"  {dur[i]:.0f}  {sz[i].astype(int).tolist()}")
print("fastest 3:")
for i in order[:3]:
    print(f"  {dur[i]:.0f}  {sz[i].astype(int).tolist()}")

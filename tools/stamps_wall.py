#!/usr/bin/env python3
"""Experiment (runs on the GPU box, needs the libalacgpu_wc.so hack build): global timeline of the cfg2 launch from the constant
100 MHz counter: first workgroup start, last entropy-wave end, against the launch duration from events."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import alac.net_amd as pkg
from alac.net_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = synth.make_config_batch(2, n_packets=n)
dev = torch.device("cuda", 0)
slot, nb = int(b["slot_ints"]), int(b["blob"].size)
d_blob = torch.zeros((nb + 63) // 16 * 16 + 64, dtype=torch.uint8, device=dev); d_blob[:nb] = torch.from_numpy(b["blob"]).to(dev)
d_off = torch.from_numpy(b["offsets"].astype(np.int64)).to(dev); d_sz = torch.from_numpy(b["sizes"].astype(np.int32)).to(dev)
d_pcm = torch.zeros((n, slot), dtype=torch.int32, device=dev)
d_ob = torch.zeros(n, dtype=torch.int32, device=dev); d_os = torch.zeros_like(d_ob); d_st = torch.zeros_like(d_ob)
nwg = (n + 7) // 8
d_stamps = torch.zeros(8 * nwg, dtype=torch.int64, device=dev)
L = pkg.lib(); fn = L.alacgpu_dbg_decode_batch_device_stamps; fn.restype = C.c_int
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
    for rep in range(3):
        d_stamps.zero_()
        assert fn(ctx._ctx, vp(d_blob), C.c_uint64(nb), vp(d_off), vp(d_sz), None, C.c_uint32(n), vp(d_pcm), C.c_uint32(slot), vp(d_ob), vp(d_os), vp(d_st), None, vp(d_stamps)) == 0
        torch.cuda.synchronize()
    ms = ctx.last_kernel_ms()
s = d_stamps.cpu().numpy().astype(np.uint64).reshape(-1, 8)
ws, we = s[:, 5].astype(np.int64), s[:, 6].astype(np.int64)
cyc = (s[:, 2] - s[:, 0]).astype(np.float64); wall = (we - ws).astype(np.float64) * 10.0   # ns
print(f"launch (events) {ms*1000:.1f} us; workgroup windows: first start .. last end = {(we.max()-ws.min())*0.01:.1f} us; starts spread {(ws.max()-ws.min())*0.01:.1f} us; "
      f"ends spread {(we.max()-we.min())*0.01:.1f} us; clock {np.median(cyc/wall):.3f} GHz; longest window {wall.max()/1000:.1f} us, median {np.median(wall)/1000:.1f} us")

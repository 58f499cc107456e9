import io, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from alac.net_amd import container, synth
from alac.net_amd.synth import m4a
import alac.net_amd as pkg
n_packets = 2584
d, sig, cfgs, _ = synth.config_descs(1, n_packets)
b = synth.make_batch(d, sig)
packets = [bytes(b["blob"][int(o):int(o) + int(s)]) for o, s in zip(b["offsets"], b["sizes"])]
data = m4a.write_m4a(packets, [4096] * n_packets)
for rep in range(3):
    t0 = time.perf_counter()
    ctx = container.AlacContext(io.BytesIO(data), batch_packets=n_packets)
    t1 = time.perf_counter()
    r = ctx.ReadBatch()
    t2 = time.perf_counter()
    r2 = None
    ctx.Dispose()
    t3 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.2f} ms  ReadBatch {1e3*(t2-t1):.2f} ms  dispose {1e3*(t3-t2):.2f} ms", flush=True)
for chunks in ("1", "2"):
    os.environ["ALACGPU_HOST_CHUNKS"] = chunks
    with pkg.AlacGpuContext(cfgs) as c:
        slot = 8200
        out = np.zeros((n_packets, slot), np.int32)
        for rep in range(3):
            t = time.perf_counter(); c.decode_batch(b["blob"], b["offsets"], b["sizes"], None, slot, out=out); print(f"chunks {chunks} decode_batch reused out: {1e3*(time.perf_counter()-t):.2f} ms")
        t = time.perf_counter(); c.decode_batch(b["blob"], b["offsets"], b["sizes"], None, slot); print(f"chunks {chunks} decode_batch fresh out: {1e3*(time.perf_counter()-t):.2f} ms")

import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import alac.net_amd as pkg
from alac.net_amd import synth
b = synth.make_config_batch(2, n_packets=4096)
dev = torch.device("cuda", 0)
n, slot, nb = 4096, int(b["slot_ints"]), int(b["blob"].size)
d_blob = torch.zeros((nb + 63) // 16 * 16 + 64, dtype=torch.uint8, device=dev); d_blob[:nb] = torch.from_numpy(b["blob"]).to(dev)
d_pcm = torch.zeros((n, slot), dtype=torch.int32, device=dev)
d_ob = torch.zeros(n, dtype=torch.int32, device=dev); d_os = torch.zeros_like(d_ob); d_st = torch.zeros_like(d_ob)
def run(order, label):
    d_off = torch.from_numpy(b["offsets"][order].astype(np.int64)).to(dev); d_sz = torch.from_numpy(b["sizes"][order].astype(np.int32)).to(dev)
    with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
        s = torch.cuda.current_stream()
        ts = []
        for rep in range(30):
            ctx.decode_batch_device(d_blob, nb, d_off, d_sz, None, n, d_pcm, slot, d_ob, d_os, d_st, stream=s.cuda_stream)
            ts.append(ctx.last_kernel_ms())
        print(f"{label:<40s} kernel {np.mean(ts[5:]):.4f} ms (min {np.min(ts[5:]):.4f})  ok={bool((d_st.cpu().numpy() == 0).all())}")
ident = np.arange(n)
rank = np.argsort(b["sizes"], kind="stable")            # small .. big
deal = rank.reshape(8, 512).T.reshape(-1)                # workgroup g gets ranks g, g+512, ...: one packet of every size octile
grouped = rank                                           # similar sizes together
rng = np.random.default_rng(1)
for rep in range(2):
    run(ident, "batch order")
    run(deal, "dealt by size rank (one per octile)")
    run(grouped, "sorted by size (similar together)")
    run(rng.permutation(n), "random order")

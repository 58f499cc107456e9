#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: decoded PCM Msamples/s on batched frames.

A "step" is one pass of the hot path (alac_decode_packets_kernel through the C ABI's
alacgpu_decode_batch_device) over one batch of synthetic packets already resident in HBM.
N=1 workload = BASELINE configs[1]: 4 096 synthetic 16-bit stereo packets, 4096 samples/frame,
LPC order 8.  N>1: one process per GPU, every rank decodes its own 4 096-packet shard
(weak scaling, no data-path collective inside the timed region); the decoded-PCM all-gather
north_star names is run and timed separately after the timed region (`allgather_ms`).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config number (2..5)")
    ap.add_argument("--packets", type=int, default=None, help="packets per GPU (default: the config's batch)")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 auto, 1 fused, 2/3/4 split, 5 two-pass)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 path)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    distributed = world > 1
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where collective operands live
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import alac.net_amd as pkg
    from alac.net_amd import synth

    per_gpu_default = {2: 4096, 3: 8192, 4: 8192, 5: 4096}  # cfg4/cfg5 are quoted per 8 GPUs: 65536/8, 32768/8
    n_packets = args.packets or per_gpu_default[args.config]

    # ---- synthetic input for this rank's shard (seeded by global packet index) ----
    t0 = time.time()
    b = synth.make_config_batch(args.config, n_packets=n_packets, first_index=rank * n_packets, want_pcm=False)
    gen_s = time.time() - t0
    descs = b["descs"]
    slot = int(b["slot_ints"])
    blob_np = b["blob"]
    blob_bytes = int(blob_np.size)
    # device allocation readable up to align_up(blob_bytes, 16)
    d_blob = torch.zeros((blob_bytes + 63) // 16 * 16 + 64, dtype=torch.uint8, device=dev)
    d_blob[:blob_bytes] = torch.from_numpy(blob_np).to(dev)
    d_off = torch.from_numpy(b["offsets"].astype(np.int64)).to(dev)  # same bits as uint64
    d_sz = torch.from_numpy(b["sizes"].astype(np.int32)).to(dev)     # same bits as uint32
    d_ci = None
    if b["cfg_idx"] is not None:
        d_ci = torch.from_numpy(b["cfg_idx"].astype(np.int16)).to(dev)
    d_pcm = torch.zeros((n_packets, slot), dtype=torch.int32, device=dev)
    d_ob = torch.zeros(n_packets, dtype=torch.int32, device=dev)
    d_os = torch.zeros(n_packets, dtype=torch.int32, device=dev)
    d_st = torch.full((n_packets,), -1, dtype=torch.int32, device=dev)

    ctx = pkg.AlacGpuContext(b["stream_cfgs"], device=local_rank)
    ctx.set_kernel_variant(args.variant)
    stream = torch.cuda.current_stream(dev)

    def step():
        ctx.decode_batch_device(d_blob, blob_bytes, d_off, d_sz, d_ci, n_packets, d_pcm, slot, d_ob, d_os, d_st,
                                stream=stream.cuda_stream)

    nch = 1 + descs["stereo"].astype(np.int64)
    samples_per_step = int((descs["n"].astype(np.int64) * nch).sum())          # S_ch = sum n*channels
    out_nc = np.array([b["stream_cfgs"][0 if b["cfg_idx"] is None else int(i)][5] for i in
                       (np.zeros(n_packets, dtype=int) if b["cfg_idx"] is None else b["cfg_idx"])], dtype=np.int64)
    algo_bytes = int(b["sizes"].astype(np.int64).sum() + (4 * descs["n"].astype(np.int64) * out_nc).sum())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record(stream)
        step()
        e1.record(stream)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        k = torch.tensor([kernel_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        kernel_ms = float(k.item())
        tot = torch.tensor([samples_per_step, algo_bytes], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        total_samples_per_step = int(tot[0].item())
    else:
        total_samples_per_step = samples_per_step

    # ---- decoded-PCM all-gather over RCCL/xGMI (outside the timed region; reported separately) ----
    allgather_ms = None
    if distributed and not args.no_allgather:
        from alac.net_amd import sharding

        src = d_pcm if args.backend == "nccl" else d_pcm.cpu()
        gathered = sharding.allgather_pcm(src, world * n_packets)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            gathered = sharding.allgather_pcm(src, world * n_packets)
        torch.cuda.synchronize(dev)
        allgather_ms = (time.perf_counter() - t1) / reps * 1e3
        assert torch.equal(gathered[rank * n_packets:(rank + 1) * n_packets], src)
        del gathered

    # ---- correctness + CPU baseline (rank 0, N=1 only; the oracle is the checker, never the product) ----
    st = d_st.cpu().numpy()
    status_ok = bool((st == 0).all()) if args.config != 5 else bool((st >= 0).all())
    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import alac_oracle_py as orc

        cfgs = orc.make_cfgs(b["stream_cfgs"])
        ncores = len(os.sched_getaffinity(0))
        # bounded sample: the first `take` packets of the same batch, one thread
        take = min(n_packets, 2048)
        sub_samples = int((descs["n"][:take].astype(np.int64) * nch[:take]).sum())
        t1 = time.perf_counter()
        ref = orc.decode_batch(cfgs, blob_np, b["offsets"][:take], b["sizes"][:take],
                               None if b["cfg_idx"] is None else b["cfg_idx"][:take], slot, n_threads=1)
        cpu1_s = time.perf_counter() - t1
        t1 = time.perf_counter()
        refall = orc.decode_batch(cfgs, blob_np, b["offsets"], b["sizes"], b["cfg_idx"], slot, n_threads=ncores)
        cpuall_s = time.perf_counter() - t1
        got = d_pcm.cpu().numpy()
        okm = refall[3] == 0
        parity = bool(np.array_equal(st, refall[3]) and np.array_equal(got[okm], refall[0][okm])
                      and np.array_equal(d_ob.cpu().numpy(), refall[1]))
        cpu_baseline = {
            "value": round(sub_samples / cpu1_s / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"first {take} packets of the same batch, C restatement of AlacFile.DecodeFrame "
                      f"(C# runtime unavailable), 1 thread",
            "all_cores_value": round(samples_per_step / cpuall_s / 1e6, 3), "all_cores": ncores,
        }

    if rank == 0:
        all_mono = all(c[5] == 1 for c in b["stream_cfgs"])
        split_auto = 4 if n_packets > (10240 if all_mono else 5120) else 3
        auto = 4 if (all_mono and 10240 < n_packets <= 20480) else 5   # mirrors the library's choice (alacgpu_api.hip: launch)
        kernel_name = {1: "alac_decode_packets_kernel", 2: "alac_decode_split1_kernel", 3: "alac_decode_split2_kernel",
                       4: "alac_decode_split4_kernel", 5: "alac_decode_ab_kernel"}[args.variant or auto]
        if all_mono and (args.variant or auto) in (3, 4):
            kernel_name = kernel_name.replace("_kernel", "_mono_kernel")
        if (args.variant or auto) == 5 and args.config == 5:
            # LPC orders above 16 in (nearly) every group of 8 packets: the work is done by the 32-tap arrangement launched
            # behind the main two-pass kernel
            kernel_name = "alac_decode_ab32_kernel (behind alac_decode_ab_kernel)"
        # HBM traffic comes from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read in-process):
        # the committed measurement of the same workload + kernel, see profiles/
        traffic = None
        tfile = os.path.join(ROOT, "profiles", f"traffic_cfg{args.config}.json")
        if os.path.exists(tfile) and args.packets is None:
            tj = json.load(open(tfile))
            if tj.get("kernel") == kernel_name:
                traffic = tj["hbm_bytes_per_launch"]
        ms_per_step = elapsed / args.steps * 1e3
        value = total_samples_per_step * args.steps / elapsed / 1e6
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        names = {2: "cfg2: 4096 synthetic 16-bit stereo packets, 4096 samples/frame, LPC order 8",
                 3: "cfg3: 24-bit stereo, 8192-sample packets, LPC order 16",
                 4: "cfg4: 16-bit mono, 4096-sample packets",
                 5: "cfg5: mixed LPC order 4-31, mixed 16/24-bit"}
        line = {
            "metric": "decoded PCM Msamples/sec (batched frames)", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": names[args.config], "packets_per_gpu": n_packets,
                       "samples_per_step_per_gpu": samples_per_step, "parallelism": f"packet-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": algo_bytes},
            "cpu_baseline": cpu_baseline,
            "parity_vs_oracle": parity, "status_ok": status_ok, "allgather_ms": allgather_ms,
            "gen_seconds": round(gen_s, 2),
        }
        print(json.dumps(line), flush=True)
    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

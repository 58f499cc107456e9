#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: decoded PCM Msamples/s on batched frames.

A "step" is one pass of the hot path (alac_decode_ab_small_kernel + alac_decode_ab32_kernel through the C ABI's
alacgpu_decode_batch_device) over one batch of synthetic packets already resident in HBM.
Workload of `value` at every N = BASELINE configs[1] ("cfg2"): 4 096 synthetic 16-bit stereo packets per GPU,
4096 samples/frame, LPC order 8.  N>1: one process per GPU, every rank decodes its own shard (weak scaling, no data-path
collective inside the timed region).  At N>1 the two multi-GPU configs of BASELINE.json are measured as well and
reported under "extra_configs": cfg4 (16-bit mono, 8 192 packets per GPU) and cfg5 (mixed LPC order / bit depth,
4 096 packets per GPU), each with the decoded-PCM all-gather north_star names -- alone (`allgather_ms`) and overlapped
with the decode chunk by chunk (`decode_allgather_overlapped_ms`).

`python bench.py --gpus N` without a launcher starts the N ranks itself (torch.distributed.run as a child process,
before anything in this process touches the GPU) and relays rank 0's JSON line and the children's exit code.
Exit code: 0 only if every rank decoded with status OK and (N=1) the whole batch equals the oracle's output.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PER_GPU_PACKETS = {2: 4096, 3: 8192, 4: 8192, 5: 4096}  # cfg4/cfg5 are quoted per 8 GPUs: 65536/8, 32768/8
NAMES = {2: "cfg2: 4096 synthetic 16-bit stereo packets, 4096 samples/frame, LPC order 8",
         3: "cfg3: 24-bit stereo, 8192-sample packets, LPC order 16",
         4: "cfg4: 16-bit mono, 4096-sample packets",
         5: "cfg5: mixed LPC order 4-31, mixed 16/24-bit"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config number (2..5) behind `value`")
    ap.add_argument("--packets", type=int, default=None, help="packets per GPU (default: the config's batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="single-thread CPU baseline: run at least this long")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="N>1: skip the cfg4 / cfg5 extra measurements")
    ap.add_argument("--extra-timeout", type=int, default=240,
                    help="N>1: seconds the all-gather + extra measurements may take before the line is printed without them (0: no limit)")
    ap.add_argument("--extra-packets", type=int, default=None, help="packets per GPU of the extra configs (rehearsals)")
    ap.add_argument("--no-host-path", action="store_true", help="N=1: skip the PCIe-inclusive and latency measurements")
    ap.add_argument("--no-big-batch", action="store_true", help="N=1: skip the extra line for a 32768-packet batch of the same config")
    ap.add_argument("--no-in-flight", action="store_true", help="N=1: skip the extra line with two batches in flight (two streams)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 path)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# self-launch: nothing in here may touch the GPU (a process that has initialised HIP must not start/replace programs
# carelessly on this pool; the parent only counts devices through sysfs, spawns, waits and relays)
# ------------------------------------------------------------------------------------------------------------------
def _visible_gpu_count():
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            try:
                props = open(os.path.join(base, node, "properties")).read()
            except OSError:
                continue
            for line in props.splitlines():
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except OSError:
        return None   # no KFD: let the ranks report it
    return n


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, argv):
    ndev = _visible_gpu_count()
    if not args.same_device and ndev is not None and ndev < args.gpus:
        # sysfs says too few: ask the runtime itself, in a child (this process stays away from the GPU)
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                               text=True, timeout=300)
            ndev = max(ndev, int(r.stdout.strip().splitlines()[-1]))
        except Exception:   # noqa: BLE001
            pass
    if not args.same_device and ndev is not None and ndev < args.gpus:
        print(json.dumps({"error": f"--gpus {args.gpus} asked for, {ndev} GPU(s) visible", "n_gpus": args.gpus}), flush=True)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes needs it on this pool)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------------------------
class Workload:
    """One rank's shard of a BASELINE config, resident in HBM, with its context."""

    def __init__(self, pkg, synth, torch, np, cfgno, n_packets, rank, dev, local_rank):
        self.np, self.torch = np, torch
        t0 = time.time()
        b = synth.make_config_batch(cfgno, n_packets=n_packets, first_index=rank * n_packets, want_pcm=False)
        self.gen_s = time.time() - t0
        self.b, self.cfgno, self.n_packets, self.dev = b, cfgno, n_packets, dev
        self.slot = int(b["slot_ints"])
        self.blob_bytes = int(b["blob"].size)
        # device allocation readable up to align_up(blob_bytes, 16)
        self.d_blob = torch.zeros((self.blob_bytes + 63) // 16 * 16 + 64, dtype=torch.uint8, device=dev)
        self.d_blob[:self.blob_bytes] = torch.from_numpy(b["blob"]).to(dev)
        self.d_off = torch.from_numpy(b["offsets"].astype(np.int64)).to(dev)  # same bits as uint64
        self.d_sz = torch.from_numpy(b["sizes"].astype(np.int32)).to(dev)     # same bits as uint32
        self.d_ci = None if b["cfg_idx"] is None else torch.from_numpy(b["cfg_idx"].astype(np.int16)).to(dev)
        self.d_pcm = torch.zeros((n_packets, self.slot), dtype=torch.int32, device=dev)
        self.d_ob = torch.zeros(n_packets, dtype=torch.int32, device=dev)
        self.d_os = torch.zeros(n_packets, dtype=torch.int32, device=dev)
        self.d_st = torch.full((n_packets,), -1, dtype=torch.int32, device=dev)
        self.ctx = pkg.AlacGpuContext(b["stream_cfgs"], device=local_rank)
        descs = b["descs"]
        nch = 1 + descs["stereo"].astype(np.int64)
        self.nch = nch
        self.samples = int((descs["n"].astype(np.int64) * nch).sum())          # S_ch = sum n*channels
        self.frames = int(descs["n"].astype(np.int64).sum())                    # S_fr = sum n (sample frames)
        ci = np.zeros(n_packets, dtype=int) if b["cfg_idx"] is None else b["cfg_idx"].astype(int)
        out_nc = np.array([b["stream_cfgs"][int(i)][5] for i in ci], dtype=np.int64)
        self.algo_bytes = int(b["sizes"].astype(np.int64).sum() + (4 * descs["n"].astype(np.int64) * out_nc).sum())

    def step(self, stream, lo=0, hi=None):
        hi = self.n_packets if hi is None else hi
        self.ctx.decode_batch_device(self.d_blob, self.blob_bytes, self.d_off[lo:hi], self.d_sz[lo:hi],
                                     None if self.d_ci is None else self.d_ci[lo:hi], hi - lo, self.d_pcm[lo:hi], self.slot,
                                     self.d_ob[lo:hi], self.d_os[lo:hi], self.d_st[lo:hi], stream=stream.cuda_stream)

    def status_ok(self):
        st = self.d_st.cpu().numpy()
        return bool((st == 0).all()) if self.cfgno != 5 else bool((st >= 0).all() and (st == 0).mean() > 0.9)

    def close(self):
        self.ctx.close()


def timed_steps(torch, dist, w, steps, warmup, dev, distributed):
    """W untimed + EXACTLY K timed steps bracketed by barrier + synchronize on both sides; returns (seconds, kernel ms)."""
    stream = torch.cuda.current_stream(dev)
    for _ in range(warmup):
        w.step(stream)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record(stream)      # the launch stream IS torch's current stream here
        w.step(stream)
        e1.record(stream)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs)
    return elapsed, float(kernel_ms)


def measure_in_flight(torch, w, dev, steps, k=2):
    """The same K steps with `k` batches IN FLIGHT: consecutive steps go round-robin to k streams (each with its own output
    buffers), so that the next batch's workgroups fill the SIMDs the current one leaves idle -- at 4096 packets a launch is
    bound by the length of one packet's serial chain, not by the chip.  Never `value` (whose steps run one behind the other);
    it is how a service that decodes a stream of batches should drive the library.  Returns a dict."""
    streams = [torch.cuda.Stream(dev) for _ in range(k)]
    outs = [(torch.zeros_like(w.d_pcm), torch.zeros_like(w.d_ob), torch.zeros_like(w.d_os), torch.full_like(w.d_st, -1))
            for _ in range(k)]

    def launch(i):
        o, st = outs[i % k], streams[i % k]
        w.ctx.decode_batch_device(w.d_blob, w.blob_bytes, w.d_off, w.d_sz, w.d_ci, w.n_packets, o[0], w.slot, o[1], o[2], o[3],
                                  stream=st.cuda_stream)
    for i in range(2 * k):
        launch(i)
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i, (e0, e1) in enumerate(evs):
        e0.record(streams[i % k])
        launch(i)
        e1.record(streams[i % k])
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    same = all(bool(torch.equal(o[0], w.d_pcm)) and bool(torch.equal(o[3], w.d_st)) for o in outs)
    return {"batches_in_flight": k, "steps": steps, "ms_per_step": round(el / steps * 1e3, 4),
            "value": round(w.samples * steps / el / 1e6, 3), "unit": "Msamples/s",
            "launch_ms": round(sum(a.elapsed_time(b) for a, b in evs) / steps, 4),
            "chip_algorithmic_GBps": round(w.algo_bytes * steps / el / 1e9, 2),
            "chip_frac_of_hbm_peak": round(w.algo_bytes * steps / el / 1e9 / HBM_PEAK_GBS, 5),
            "same_output_as_value_run": same,
            "note": "steps issued round-robin to %d streams, one context; each launch takes longer (launch_ms), the chip decodes "
                    "more per second" % k}


def reduce_max(torch, dist, cdev, *vals):
    t = torch.tensor(list(vals), dtype=torch.float64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t.tolist()]


def reduce_sum(torch, dist, cdev, *vals):
    t = torch.tensor(list(vals), dtype=torch.int64, device=cdev)
    dist.all_reduce(t)
    return [int(x) for x in t.tolist()]


def kernel_source_sha():
    h = hashlib.sha256()
    for fn in ("alac_kernels.hip", "alac_device.h", "alac_kernels.h", "Makefile"):   # (the Makefile holds the scheduler flags)
        h.update(open(os.path.join(ROOT, "alac.net_amd", "csrc", fn), "rb").read())
    return h.hexdigest()[:16]


def measure_allgather(torch, dist, sharding, w, world, rank, dev, backend, reps=3):
    """(collective alone, decode + all-gather overlapped chunk by chunk), milliseconds; checks the gathered PCM."""
    n = w.n_packets
    cuda_coll = backend == "nccl"
    src = w.d_pcm if cuda_coll else w.d_pcm.cpu()
    gathered = sharding.allgather_pcm(src, world * n)
    torch.cuda.synchronize(dev)
    dist.barrier()
    t1 = time.perf_counter()
    for _ in range(reps):
        gathered = sharding.allgather_pcm(src, world * n)
    torch.cuda.synchronize(dev)
    alone_ms = (time.perf_counter() - t1) / reps * 1e3
    ok = bool(torch.equal(gathered[rank * n:(rank + 1) * n], src))
    del gathered
    # overlapped: the shard decodes in chunks; chunk k's all-gather runs while chunk k+1 decodes
    pipe = sharding.ChunkedDecodeAllGather(w.d_pcm, world, n_chunks=4, cuda_collective=cuda_coll)
    stream = torch.cuda.current_stream(dev)
    full = pipe.run(lambda lo, hi: w.step(stream, lo, hi))
    torch.cuda.synchronize(dev)
    dist.barrier()
    t1 = time.perf_counter()
    for _ in range(reps):
        full = pipe.run(lambda lo, hi: w.step(stream, lo, hi))
    torch.cuda.synchronize(dev)
    over_ms = (time.perf_counter() - t1) / reps * 1e3
    ok = ok and bool(torch.equal(full[rank * n:(rank + 1) * n], src))
    del full, pipe
    return alone_ms, over_ms, ok


def measure_allgather_native(pkg, torch, dist, np, w, world, rank, dev, cdev, reps=3):
    """The same two figures through the C ABI (alacgpu_comm_*: ncclAllGather from librccl on the caller's stream; the
    overlapped form with its own collective stream): what a native host (the C# AlacContext) gets.  Weak scaling: every
    rank's shard has n packets, global packet r*n + i = rank r's packet i.  Returns (alone ms, overlapped ms, ok)."""
    n, slot = w.n_packets, w.slot
    # every rank first finds out, by itself, whether RCCL loads through the library (an id made here and thrown away), and the
    # ranks agree on the answer BEFORE anything collective happens through the C ABI: a rank that failed alone would leave the
    # others waiting in ncclCommInitRank
    try:
        my_id, ok_here = pkg.AlacGpuComm.unique_id(), 1
    except Exception:   # noqa: BLE001
        my_id, ok_here = np.zeros(128, dtype=np.uint8), 0
    flag = torch.tensor([ok_here], dtype=torch.int32, device=cdev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        raise RuntimeError("RCCL not loadable through libalacgpu.so on some rank")
    uid = torch.from_numpy(my_id).to(cdev)
    dist.broadcast(uid, src=0)
    first = np.arange(world + 1, dtype=np.uint32) * n
    with pkg.AlacGpuComm(w.ctx, uid.cpu().numpy(), rank, world) as comm:
        full = torch.zeros((world * n, slot), dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream(dev)
        full[rank * n:(rank + 1) * n].copy_(w.d_pcm)
        comm.allgather_pcm(full, first, slot, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(reps):
            comm.allgather_pcm(full, first, slot, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        alone_ms = (time.perf_counter() - t1) / reps * 1e3
        ok = bool(torch.equal(full[rank * n:(rank + 1) * n], w.d_pcm))
        # every rank's shard must have arrived: compare a checksum of each shard with its owner's
        sums = torch.stack([full[r * n:(r + 1) * n].sum(dtype=torch.int64) for r in range(world)]).to(cdev)
        mine = sums.clone()
        dist.broadcast(sums, src=0)
        ok = ok and bool(torch.equal(sums, mine))
        # overlapped: global metadata on every GPU; this rank decodes its range straight into `full`
        def cat(t, fill=0):
            g = torch.full((world * n,) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=dev)
            g[rank * n:(rank + 1) * n].copy_(t)
            return g
        g_off, g_sz = cat(w.d_off), cat(w.d_sz)
        g_ci = None if w.d_ci is None else cat(w.d_ci)
        g_ob, g_os, g_st = cat(w.d_ob), cat(w.d_os), cat(w.d_st, -1)
        full.zero_()
        run = lambda: comm.decode_allgather_device(w.d_blob, w.blob_bytes, g_off, g_sz, g_ci, first, full, slot, g_ob, g_os, g_st,
                                                   n_chunks=4, stream=stream.cuda_stream)
        run()
        torch.cuda.synchronize(dev)
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize(dev)
        over_ms = (time.perf_counter() - t1) / reps * 1e3
        ok = ok and bool(torch.equal(full[rank * n:(rank + 1) * n], w.d_pcm)) and bool(torch.equal(g_st[rank * n:(rank + 1) * n], w.d_st))
        sums2 = torch.stack([full[r * n:(r + 1) * n].sum(dtype=torch.int64) for r in range(world)]).to(cdev)
        ok = ok and bool(torch.equal(sums2, mine))
        del full
    return alone_ms, over_ms, ok


def gather_figures(pkg, torch, dist, sharding, np, w, world, rank, dev, cdev, backend):
    """(alone ms, overlapped ms, ok, which): through the C ABI's RCCL entry points when the collective operands can stay in
    HBM (backend nccl), through torch.distributed otherwise (gloo rehearsals) -- or when the native path fails, which the
    line then says."""
    if backend == "nccl":
        err = None
        try:
            a, o, ok = measure_allgather_native(pkg, torch, dist, np, w, world, rank, dev, cdev)
        except Exception as e:   # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        # all ranks take the same road from here on
        bad = torch.tensor([1 if err else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()) == 0:
            return a, o, ok, "alacgpu_comm (C ABI, RCCL)"
        note = f"alacgpu_comm failed ({err or 'on another rank'}); torch.distributed instead"[:300]
        a, o, ok = measure_allgather(torch, dist, sharding, w, world, rank, dev, backend)
        return a, o, ok, note
    a, o, ok = measure_allgather(torch, dist, sharding, w, world, rank, dev, backend)
    return a, o, ok, f"torch.distributed ({backend})"


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        sys.exit(spawn_ranks(args, argv))   # the children are the ranks; this process never touches the GPU
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    ndev = torch.cuda.device_count()       # (does not initialise the GPU)
    if ndev <= local_rank:
        if rank == 0:
            print(json.dumps({"error": f"rank needs cuda:{local_rank}, {ndev} device(s) visible", "n_gpus": world}), flush=True)
        raise SystemExit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where collective operands live
    dist = None
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import alac.net_amd as pkg
    from alac.net_amd import sharding, synth

    n_packets = args.packets or PER_GPU_PACKETS[args.config]
    w = Workload(pkg, synth, torch, np, args.config, n_packets, rank, dev, local_rank)
    elapsed, kernel_ms = timed_steps(torch, dist, w, args.steps, args.warmup, dev, distributed)
    # (right behind the timed region, before this process makes any other stream: the runtime maps streams onto a handful of
    # hardware queues in creation order, and two streams that share a queue do not overlap)
    in_flight = None
    if rank == 0 and world == 1 and not args.no_in_flight:
        try:
            in_flight = measure_in_flight(torch, w, dev, max(args.steps, 8))
        except Exception as e:   # noqa: BLE001
            in_flight = {"error": f"{type(e).__name__}: {e}"[:300]}

    status_ok = w.status_ok()
    total_samples = w.samples
    if distributed:
        elapsed, kernel_ms = reduce_max(torch, dist, cdev, elapsed, kernel_ms)
        total_samples, bad = reduce_sum(torch, dist, cdev, w.samples, 0 if status_ok else 1)
        status_ok = bad == 0

    # ---- N>1: decoded-PCM all-gather (outside the timed region) and the two multi-GPU configs of BASELINE.json ----
    allgather_ms = overlapped_ms = None
    gather_ok = True
    gather_via = None
    extra = {}
    extra_error = None
    cpu_baseline = parity = host_path = big_batch = None

    def emit():
        """rank 0's JSON line from whatever has been measured so far."""
        # the main kernel's build by batch size (alacgpu_api.hip: launch): 16-step units up to 4096 packets, 8-step units with 128
        # registers up to 10240, with 96 registers up to 12288, 16 packets per workgroup above
        first = ("alac_decode_ab_small_kernel" if n_packets <= 4096 else "alac_decode_ab_kernel" if n_packets <= 10240
                 else "alac_decode_ab5_kernel" if n_packets <= 12288 else "alac_decode_ab_dense_kernel")
        kernel_name = first
        if args.config == 3 and first != "alac_decode_ab_dense_kernel":
            # LPC order 16: two taps per lane of the FIR wave: the second launch (the dense arrangement keeps such groups)
            kernel_name = f"alac_decode_ab32_kernel (two taps per lane; behind {first})"
        if args.config == 5:
            # LPC orders above 16 in (nearly) every group of 8 packets: four taps per lane, the second launch
            kernel_name = f"alac_decode_ab32_kernel (four taps per lane; behind {first})"
        # HBM traffic comes from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read in-process): the committed
        # measurement of the same workload, valid only for the kernel sources it was taken with
        traffic, traffic_note, issue = None, None, None
        tfile = os.path.join(ROOT, "profiles", f"traffic_cfg{args.config}.json")
        if os.path.exists(tfile) and args.packets is None:
            tj = json.load(open(tfile))
            if tj.get("kernel_source_sha") == kernel_source_sha():
                traffic = tj["hbm_bytes_per_launch"]
                # the issue-rate ceiling that makes the HBM fraction interpretable.  The kernels are bound
                # by VALU issue (a SIMD issues one wave64 VALU instruction per 4 cycles), not by bandwidth: ceiling = SIMDs x clock
                # / 4 / (VALU wave-instructions per sample, from the committed SQ_INSTS_VALU pass of this very workload)
                if tj.get("valu_wave_instr_per_launch"):
                    ips = tj["valu_wave_instr_per_launch"] / w.samples
                    simds, clock = 1024, float(tj.get("effective_clock_GHz") or 2.4)
                    ceiling = simds * clock * 1e3 / 4.0 / ips          # Msamples/s
                    issue = {"valu_wave_instr_per_sample": round(ips, 3), "simd_count": simds, "clock_ghz": round(clock, 3),
                             "cycles_per_wave_instr": 4, "ceiling_msamples": round(ceiling, 1),
                             "source": tj.get("source")}
            else:
                traffic_note = "profiles/traffic_cfg%d.json was measured with other kernel sources (stale): not reported" % args.config
        ms_per_step = elapsed / args.steps * 1e3
        value = total_samples * args.steps / elapsed / 1e6
        achieved = w.algo_bytes / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "decoded PCM Msamples/sec (batched frames)", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": NAMES[args.config], "packets_per_gpu": n_packets,
                       "samples_per_step_per_gpu": w.samples, "parallelism": f"packet-sharded x{world}"},
            "value_frames": round(value * w.frames / w.samples, 3),    # S_fr: sample frames per second
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": w.algo_bytes},
            "cpu_baseline": cpu_baseline,
            "parity_vs_oracle": parity, "status_ok": status_ok,
            "allgather_ms": None if allgather_ms is None else round(allgather_ms, 4),
            "decode_allgather_overlapped_ms": None if overlapped_ms is None else round(overlapped_ms, 4),
            "allgather_ok": gather_ok if distributed else None, "allgather_via": gather_via,
            "gen_seconds": round(w.gen_s, 2),
        }
        if traffic_note:
            line["roofline"]["traffic_note"] = traffic_note
        if issue:
            # per GPU, kernel time only (like `frac`): what the decode itself reaches of the chip's VALU issue rate
            issue["frac_of_issue_ceiling"] = round(w.samples / (kernel_ms * 1e-3) / 1e6 / issue["ceiling_msamples"], 4)
            line["roofline"]["issue_ceiling"] = issue
        if extra:
            line["extra_configs"] = extra
        if extra_error:
            line["extra_error"] = extra_error
        if host_path:
            line["host_path"] = host_path
        if big_batch:
            line["big_batch"] = big_batch
        if in_flight:
            line["two_in_flight"] = in_flight
        print(json.dumps(line), flush=True)

    # a collective that never returns (a fabric or library fault on a node this code has never met) must not take the headline
    # with it: past --extra-timeout seconds rank 0 prints the line it has and every rank leaves
    watchdog = None
    if distributed and args.extra_timeout > 0:
        import threading

        def give_up():
            nonlocal extra_error
            extra_error = f"the multi-GPU extras did not finish within {args.extra_timeout} s; line printed without them"
            if rank == 0:
                emit()
            os._exit(3)
        watchdog = threading.Timer(args.extra_timeout + (0 if rank == 0 else 15), give_up)
        watchdog.daemon = True
        watchdog.start()
    # (the headline measurement above is complete at this point; a failure below -- out of memory, a collective that the
    # node's fabric refuses -- is reported in the line as "extra_error" instead of taking `value` down with it)
    try:
        if distributed and not args.no_allgather:
            allgather_ms, overlapped_ms, gather_ok, gather_via = gather_figures(pkg, torch, dist, sharding, np, w, world, rank, dev, cdev,
                                                                                 args.backend)
        if distributed and not args.no_extra:
            for cfgno in (4, 5):
                if cfgno == args.config:
                    continue
                npk = args.extra_packets or PER_GPU_PACKETS[cfgno]
                we = Workload(pkg, synth, torch, np, cfgno, npk, rank, dev, local_rank)
                ksteps = max(1, min(args.steps, 10))
                el, kms = timed_steps(torch, dist, we, ksteps, min(args.warmup, 2), dev, True)
                ok = we.status_ok()
                el, kms = reduce_max(torch, dist, cdev, el, kms)
                tot, ab, bad = reduce_sum(torch, dist, cdev, we.samples, we.algo_bytes, 0 if ok else 1)
                row = {"workload": NAMES[cfgno], "packets_per_gpu": npk, "steps": ksteps,
                       "value": round(tot * ksteps / el / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(el / ksteps * 1e3, 4),
                       "kernel_ms": round(kms, 4), "roofline_frac": round(we.algo_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                       "status_ok": bad == 0}
                if not args.no_allgather:
                    a_ms, o_ms, g_ok, _ = gather_figures(pkg, torch, dist, sharding, np, we, world, rank, dev, cdev, args.backend)
                    a_ms, o_ms = reduce_max(torch, dist, cdev, a_ms, o_ms)
                    row.update(allgather_ms=round(a_ms, 4), decode_allgather_overlapped_ms=round(o_ms, 4),
                               allgather_bytes_per_rank=int(we.d_pcm.numel() * 4))
                    gather_ok = gather_ok and g_ok
                status_ok = status_ok and bad == 0
                extra[f"cfg{cfgno}"] = row
                we.close()
                del we
    except Exception as e:   # noqa: BLE001
        extra_error = f"{type(e).__name__}: {e}"[:400]
    if distributed and extra_error is None:
        try:
            (gbad,) = reduce_sum(torch, dist, cdev, 0 if gather_ok else 1)
            gather_ok = gbad == 0
            if allgather_ms is not None:
                allgather_ms, overlapped_ms = reduce_max(torch, dist, cdev, allgather_ms, overlapped_ms)
        except Exception as e:   # noqa: BLE001
            extra_error = f"{type(e).__name__}: {e}"[:400]

    # ---- correctness + CPU baseline (rank 0, N=1 only; the oracle is the checker, never the product) ----
    b = w.b
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import alac_oracle_py as orc

        cfgs = orc.make_cfgs(b["stream_cfgs"])
        allcores = sorted(os.sched_getaffinity(0))
        ncores = len(allcores)
        # whole batch on all cores: the parity reference
        t1 = time.perf_counter()
        refall = orc.decode_batch(cfgs, b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], w.slot, n_threads=ncores)
        cpuall_s = time.perf_counter() - t1
        got = w.d_pcm.cpu().numpy()
        st = w.d_st.cpu().numpy()
        # every packet's own n * channels ints (what a slot holds beyond them is scratch by contract, include/alacgpu.h)
        ci_h = np.zeros(n_packets, dtype=np.int64) if b["cfg_idx"] is None else b["cfg_idx"].astype(np.int64)
        nc_h = np.array([int(c[5]) for c in b["stream_cfgs"]], dtype=np.int64)[ci_h]
        cnt_h = np.where(refall[3] == 0, refall[2].astype(np.int64) * nc_h, 0)
        mask = np.arange(w.slot, dtype=np.int64)[None, :] < cnt_h[:, None]
        parity = bool(np.array_equal(st, refall[3]) and np.array_equal(got[mask], refall[0][mask])
                      and np.array_equal(w.d_ob.cpu().numpy(), refall[1]) and np.array_equal(w.d_os.cpu().numpy(), refall[2]))
        # single thread, pinned to one core, the same batch over and over for >= --cpu-seconds
        os.sched_setaffinity(0, {allcores[len(allcores) // 2]})
        try:
            reps, t1 = 0, time.perf_counter()
            while True:
                orc.decode_batch(cfgs, b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], w.slot, n_threads=1)
                reps += 1
                cpu1_s = time.perf_counter() - t1
                if cpu1_s >= args.cpu_seconds:
                    break
        finally:
            os.sched_setaffinity(0, set(allcores))
        cpu_baseline = {
            "value": round(w.samples * reps / cpu1_s / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"the same {n_packets}-packet batch decoded {reps} times ({cpu1_s:.1f} s) by the C restatement of "
                      f"AlacFile.DecodeFrame (oracle/alac_oracle.c; C# runtime unavailable), 1 thread pinned to one core",
            "all_cores_value": round(w.samples / cpuall_s / 1e6, 3), "all_cores": ncores,
        }
    if rank == 0 and world == 1 and not args.no_host_path:
        host_path = measure_host_path(pkg, np, w)
    # ---- the same config in a batch big enough to fill the chip (never `value`: BASELINE quotes the metric on 4096 packets) ----
    if rank == 0 and world == 1 and not args.no_big_batch and args.packets is None and args.config == 2:
        try:
            wb = Workload(pkg, synth, torch, np, args.config, 32768, 0, dev, local_rank)
            el, kms = timed_steps(torch, dist, wb, 10, 3, dev, False)
            big_batch = {"packets": 32768, "steps": 10, "value": round(wb.samples * 10 / el / 1e6, 3), "unit": "Msamples/s",
                         "kernel": "alac_decode_ab_dense_kernel", "kernel_ms": round(kms, 4),
                         "roofline_frac": round(wb.algo_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "status_ok": wb.status_ok(),
                         "note": "the main kernel's 16-packets-per-workgroup arrangement (batches above 12288 packets); "
                                 "bound by VALU issue, not by one packet's serial chain"}
            wb.close()
            del wb
        except Exception as e:   # noqa: BLE001
            big_batch = {"error": f"{type(e).__name__}: {e}"[:300]}

    if watchdog:
        watchdog.cancel()
    if rank == 0:
        emit()
    w.close()
    if distributed:
        try:
            dist.destroy_process_group()
        except Exception:   # noqa: BLE001
            pass
    # A decode that is wrong takes the run down.  A gather that is wrong or did not work (the collectives have only ever met a
    # two-rank rehearsal on one GPU) is outside `value`: the line says so ("allgather_ok": false, "extra_error"), the exit
    # code stays 0, so that a scaling run keeps its per-N decode figures.
    if not status_ok or parity is False:
        raise SystemExit(1)


def measure_host_path(pkg, np, w):
    """PCIe-inclusive figures (never `value`): alacgpu_decode_batch on host buffers -- upload, decode and download
    overlapped range by range -- with reused pinned buffers and with ordinary (pageable) ones; and the latency of small
    calls (alacgpu_decode_frame = one packet; batches of 1 / 8 / 64)."""
    b = w.b
    n = w.n_packets
    out = {}
    ctx = w.ctx

    def bench(fn, reps):
        fn()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t) / reps * 1e3

    with pkg.PinnedBuffer((n, w.slot), np.int32) as ppcm, pkg.PinnedBuffer(b["blob"].size, np.uint8) as pblob:
        pblob.array[:] = b["blob"]
        for fmt, name in ((0, "int32"), (1, "packed")):
            ctx.set_output_format(fmt)
            ms = bench(lambda: ctx.decode_batch(pblob.array, b["offsets"], b["sizes"], b["cfg_idx"], w.slot, out=ppcm.array), 5)
            out[f"{name}_pinned_ms"] = round(ms, 3)
            out[f"{name}_pinned_msamples_per_s"] = round(w.samples / ms / 1e3, 1)
        pcm = np.zeros((n, w.slot), dtype=np.int32)
        for fmt, name in ((0, "int32"), (1, "packed")):
            ctx.set_output_format(fmt)
            ms = bench(lambda: ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], w.slot, out=pcm), 5)
            out[f"{name}_pageable_ms"] = round(ms, 3)
        ctx.set_output_format(0)
    # small calls
    o0, s0 = int(b["offsets"][0]), int(b["sizes"][0])
    pkt = b["blob"][o0:o0 + s0].copy()
    ci0 = 0 if b["cfg_idx"] is None else int(b["cfg_idx"][0])
    out["decode_frame_ms"] = round(bench(lambda: ctx.decode_frame(ci0, pkt), 20), 4)
    for k in (1, 8, 64):
        if k > n:
            break
        sl = np.zeros((k, w.slot), dtype=np.int32)
        ci = None if b["cfg_idx"] is None else b["cfg_idx"][:k]
        out[f"batch{k}_ms"] = round(bench(lambda: ctx.decode_batch(b["blob"], b["offsets"][:k], b["sizes"][:k], ci, w.slot, out=sl), 20), 4)
    out["note"] = ("host buffers in, host buffers out (PCIe inclusive); decode_frame_ms = one packet through "
                   "alacgpu_decode_frame; a serial chain of 2 x n symbols bounds a lone packet at about 0.7 ms")
    return out


if __name__ == "__main__":
    main()

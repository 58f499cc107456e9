#!/usr/bin/env python3
"""BASELINE config 1: a 16-bit stereo 44.1 kHz .m4a decoded through the AlacContext surface.

Builds a synthetic 4-minute file (2584 packets x 4096 frames; the reference ships no audio), then
  * GPU path: alac.net_amd.container.AlacContext -- demux, ReadBatch (one GPU batch per K packets), and the
    reference's playback loop `while ((n = ctx.Read(buf)) > 0)`;
  * CPU port: the oracle's DecodeFrame + FormatSamples, one packet per call, one thread (what the reference's
    AlacContext.Read does per packet; checker/baseline only).
Prints one JSON line with Msamples/s and x real-time for each, after checking that the bytes are identical.
"""
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]

import numpy as np

from alac.net_amd import container, synth
from alac.net_amd.synth import m4a
import alac_oracle_py as orc


def main():
    n_packets = 2584
    d, sig, cfgs, _ = synth.config_descs(1, n_packets)
    b = synth.make_batch(d, sig)
    packets = [bytes(b["blob"][int(o):int(o) + int(s)]) for o, s in zip(b["offsets"], b["sizes"])]
    data = m4a.write_m4a(packets, [4096] * n_packets)
    total_frames = 4096 * n_packets
    seconds_of_audio = total_frames / 44100.0

    # ---- GPU path through the AlacContext mirror ----
    def run_gpu(batch_packets):
        buf = np.zeros(1024 * 80, dtype=np.uint8)
        out = bytearray()
        t = time.perf_counter()
        with container.AlacContext(io.BytesIO(data), batch_packets=batch_packets) as ctx:
            while True:
                n = ctx.Read(buf)
                if n <= 0:
                    break
                out += bytes(buf[:n])
        return time.perf_counter() - t, bytes(out)

    run_gpu(256)  # warm-up (library load, first launch)
    t_read, pcm_gpu = run_gpu(512)

    def run_gpu_batch(batch_packets):
        t = time.perf_counter()
        frames = 0
        with container.AlacContext(io.BytesIO(data), batch_packets=batch_packets) as ctx:
            while True:
                r = ctx.ReadBatch()
                if r is None:
                    break
                frames += int(r[2].sum())
        return time.perf_counter() - t, frames

    t_batch, frames = run_gpu_batch(n_packets)
    assert frames == total_frames

    # ---- CPU port: one packet per call, one thread ----
    cfg = orc.make_cfgs(cfgs)
    t = time.perf_counter()
    out = bytearray()
    for p in packets:
        st, pcm, ob, n = orc.decode_frame(cfg, p, capacity=4096 * 2 + 8)
        out += orc.format_samples(2, pcm, ob).tobytes()
    t_cpu = time.perf_counter() - t
    assert bytes(out) == pcm_gpu, "GPU AlacContext.Read bytes differ from the CPU port"

    samples = total_frames * 2
    print(json.dumps({
        "config": "cfg1: 16-bit stereo 44.1 kHz .m4a (synthetic, 2584 packets = %.0f s of audio) via AlacContext" % seconds_of_audio,
        "gpu_read_loop": {"seconds": round(t_read, 4), "Msamples_per_s": round(samples / t_read / 1e6, 1),
                          "x_realtime": round(seconds_of_audio / t_read), "batch_packets": 512,
                          "note": "demux + GPU batches + per-packet Read() in Python (host loop dominates)"},
        "gpu_read_batch": {"seconds": round(t_batch, 4), "Msamples_per_s": round(samples / t_batch / 1e6, 1),
                           "x_realtime": round(seconds_of_audio / t_batch), "note": "demux + one ReadBatch for the whole file, PCIe inclusive"},
        "cpu_port_1thread": {"seconds": round(t_cpu, 4), "Msamples_per_s": round(samples / t_cpu / 1e6, 1),
                             "x_realtime": round(seconds_of_audio / t_cpu)},
        "bytes_identical": True,
    }))


if __name__ == "__main__":
    main()

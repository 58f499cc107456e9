#!/usr/bin/env python3
"""Generates tests/golden/golden_v3.npz: BASELINE-shaped, FULL-LENGTH packets and the outputs the LITERAL restatement of
AlacFile.cs (oracle/alacfile_literal.py) gives for them (golden_v2's packets are at most 256
samples long; coefficient drift over thousands of steps, history saturation and many ring wraps were pinned by the C
oracle alone).

Packets (all from the synthetic encoder's BASELINE batches, i.e. exactly what bench.py decodes):
  cfg2  16-bit stereo, n 4096, order 8, mix weight 1            cfg4  16-bit mono, n 4096, order 8
  cfg3  24-bit stereo, n 8192, order 16, ub 0 and ub 1          cfg5  order 31; order >= 29 at 24 bits; a short `hassize`
  a loud 16-bit packet whose history saturates (App. B Q19)           packet; an uncompressed one
  rice_kmodifier 20 and 40 (AlacFile.cs:82 takes any byte; the run-length mask is (1 << kb) - 1 with C#'s five-bit shift count)
  a one-channel element with prediction type 2 decoded by a decoder that has decoded a normal frame before: the reference
  hands out _outputsamplesBufferA, which then IS the residual buffer (AlacFile.cs:484-496 with :486)
The reference (teekay/ALAC.NET) holds no fixtures and cannot run here (C#, no .NET): parity stays "unpinned" by
reference-held data; this file widens what the second, independent reading covers.
Build-container only (pure Python, a few minutes):  python tests/golden/make_golden_v3.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

from alac.net_amd import synth  # noqa: E402
import alacfile_literal as lit  # noqa: E402


def packet(b, p):
    o, s = int(b["offsets"][p]), int(b["sizes"][p])
    return bytes(b["blob"][o:o + s])


def cfg_of(b, p):
    ci = 0 if b["cfg_idx"] is None else int(b["cfg_idx"][p])
    return tuple(int(x) for x in b["stream_cfgs"][ci])


def literal_decoder(cfg):
    frame_len, sample_size, pb, mb, kb, nch = cfg
    f = lit.AlacFile(sample_size, nch)
    cd = [0] * 24 + [(frame_len >> 24) & 255, (frame_len >> 16) & 255, (frame_len >> 8) & 255, frame_len & 255, 0, sample_size, pb,
                     mb, kb, nch, 0, 255, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0xAC, 0x44]
    f.SetInfo(cd)
    return f


def run(f, pk):
    buf = bytearray(1024 * 80)
    buf[:len(pk)] = pk
    out = lit.new_int_array(1024 * 80)
    ret = f.DecodeFrame(buf, out)
    return out, ret


def main():
    items = []   # (label, cfg, packet bytes, decoder-state: None = fresh, else a packet to decode first)
    b2 = synth.make_config_batch(2, n_packets=8, n_threads=1)
    items.append(("cfg2 n4096 order 8 weight 1", cfg_of(b2, 0), packet(b2, 0), None))
    b3 = synth.make_config_batch(3, n_packets=8, n_threads=1)
    items.append(("cfg3 n8192 order 16 ub0", cfg_of(b3, 0), packet(b3, 0), None))
    items.append(("cfg3 n8192 order 16 ub1", cfg_of(b3, 1), packet(b3, 1), None))
    b4 = synth.make_config_batch(4, n_packets=8, n_threads=1)
    items.append(("cfg4 mono n4096 order 8", cfg_of(b4, 0), packet(b4, 0), None))
    b5 = synth.make_config_batch(5, n_packets=1024, n_threads=2)
    d = b5["descs"]
    ss = np.array([cfg_of(b5, p)[1] for p in range(1024)])
    full = (d["n"] == 4096) & (d["escape"] == 0)
    pick = {
        "cfg5 order 31 (delta mode)": np.nonzero(full & ((d["pred_order"][:, 0] == 31) | (d["pred_order"][:, 1] == 31)))[0],
        "cfg5 24-bit order >= 29": np.nonzero(full & (ss == 24) & (d["pred_order"].max(axis=1) >= 29) & (d["pred_order"].max(axis=1) <= 30))[0],
        "cfg5 hassize short": np.nonzero((d["n"] < 4096) & (d["n"] > 1500) & (d["escape"] == 0))[0],
        "cfg5 uncompressed": np.nonzero(d["escape"] == 1)[0],
    }
    for label, idx in pick.items():
        assert len(idx), label
        p = int(idx[0])
        items.append((f"{label} (packet {p})", cfg_of(b5, p), packet(b5, p), None))
    # history saturation over a whole frame (Q19): loud, noisy 16-bit stereo
    dl = synth.packet_descs(1, n=4096, max_samples_per_frame=4096)
    loud = synth.default_signal(0x10AD)
    loud["amp_lo_log2"], loud["amp_hi_log2"], loud["noise_sigma"] = 14.5, 15.0, 12000.0
    bl = synth.make_batch(dl, loud, want_pcm=False, n_threads=1)
    items.append(("loud n4096: escapes, history saturation", (4096, 16, 40, 10, 14, 2), packet(bl, 0), None))
    # rice_kmodifier beyond 16: silence-heavy content (the run-length symbols are what the mask touches)
    for kb in (20, 40):
        dk = synth.packet_descs(1, n=1024, max_samples_per_frame=4096)
        dk["rice_kmodifier"][0] = kb
        sig = synth.default_signal(0x6B00 + kb)
        sig["silence_prob"] = 1.0
        sig["silence_min"], sig["silence_max"] = 16, 300
        bk = synth.make_batch(dk, sig, want_pcm=False, n_threads=1)
        items.append((f"rice_kmodifier {kb}", (4096, 16, 40, 10, kb, 2), packet(bk, 0), None))
    # one-channel element, prediction type 2, behind a normal frame on the same decoder
    dm = synth.packet_descs(2, n=512, max_samples_per_frame=4096, stereo=0)
    dm["pred_type"][1] = [2, 0]
    bm = synth.make_batch(dm, synth.default_signal(0x9D7), want_pcm=False, n_threads=1)
    items.append(("mono prediction type 2 behind a normal frame", (4096, 16, 40, 10, 14, 1), packet(bm, 1), packet(bm, 0)))

    slot = 16384
    n_all = len(items)
    pcm = np.zeros((n_all, slot), dtype=np.int32)
    ref_ret = np.zeros(n_all, dtype=np.int64)
    n_samples = np.zeros(n_all, dtype=np.int32)
    cfgs, cfg_idx, packets, labels = [], [], [], []
    for p, (label, cfg, pk, before) in enumerate(items):
        f = literal_decoder(cfg)
        if before is not None:
            run(f, before)
        out, ret = run(f, pk)
        ref_ret[p] = ret
        n = ret // ((cfg[1] // 8) * cfg[5])
        n_samples[p] = n
        pcm[p, : n * cfg[5]] = lit.canonical_from_reference_layout(out, n, cfg[1], cfg[5])
        if cfg not in cfgs:
            cfgs.append(cfg)
        cfg_idx.append(cfgs.index(cfg))
        packets.append(pk)
        labels.append(label)
        print(f"{p}: {label}: {len(pk)} bytes, cfg {cfg}, return {ret}, n {n}", flush=True)
    sizes = np.array([len(x) for x in packets], dtype=np.uint32)
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    blob = np.frombuffer(b"".join(packets) + bytes(16), dtype=np.uint8)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v3.npz")
    np.savez_compressed(out, blob=blob, offsets=offsets, sizes=sizes, cfg_idx=np.array(cfg_idx, dtype=np.uint16),
                        cfgs=np.array(cfgs, dtype=np.int64), pcm=pcm, ref_ret=ref_ret, n_samples=n_samples, slot_ints=np.int64(slot),
                        labels=np.array(labels))
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Long randomized differential run: HIP path  vs the CPU oracle, bit for bit.
Not part of the test suite (minutes); usage: python tools/stress_parity.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]

import numpy as np

import alac.net_amd as pkg
from alac.net_amd import synth
import alac_oracle_py as orc


def recipes(rng, count, stereo, is24, orders):
    d = synth.packet_descs(count, max_samples_per_frame=4096, sample_size=24 if is24 else 16, stereo=int(stereo))
    d["n"] = rng.integers(1, 4097, count)
    d["n"][rng.random(count) < 0.5] = 4096
    d["pred_order"] = rng.choice(orders, (count, 2))
    d["quant"] = rng.integers(0, 16, (count, 2))
    d["quant"][rng.random(count) < 0.6] = 9
    d["ricemod"] = rng.integers(0, 8, (count, 2))
    d["ricemod"][rng.random(count) < 0.6] = 4
    d["mix_shift"] = rng.integers(0, 9, count)
    d["mix_weight"] = np.minimum(rng.integers(0, 256, count), 1 << d["mix_shift"].astype(np.int64))
    d["ub"] = rng.integers(0, 3 if is24 else 1, count)
    d["coef_mode"] = np.where(rng.random(count) < 0.3, 1, 0)
    d["coefs"] = rng.integers(-3000, 3000, (count, 2, 32))
    d["escape"] = rng.random(count) < 0.03
    return d


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    rounds = packets = skipped = 0
    last_note = t0
    while time.time() - t0 < budget:
        stereo, is24 = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        orders = [np.arange(0, 32), np.arange(1, 9), np.array([8]), np.arange(9, 17), np.arange(17, 32)][int(rng.integers(0, 5))]
        count = int(rng.integers(1, 97))
        big = rng.random() < 0.08     # more than 256 groups of 8: orders above 16 then take ONE FIR wave with four taps per lane
        if big:
            count = int(rng.integers(2049, 2300))
        d = recipes(rng, count, stereo, is24, orders)
        if big:
            d["n"] = np.minimum(d["n"], rng.integers(33, 700, count))
        if stereo and rng.random() < 0.3:
            d["stereo"] = rng.integers(0, 2, count)     # one-channel elements inside a two-channel stream: L = sample, R = 0
        sig = synth.default_signal(int(rng.integers(0, 1 << 31)))
        sig["silence_prob"] = float(rng.choice([0.0, 0.3, 1.0]))
        sig["silence_min"], sig["silence_max"] = 1, int(rng.choice([40, 3000]))
        if rng.random() < 0.5:
            sig["amp_lo_log2"], sig["amp_hi_log2"], sig["noise_sigma"] = 13.0, 15.0, float(rng.choice([300.0, 3000.0, 12000.0]))
        if rng.random() < 0.2:
            sig["amp_lo_log2"], sig["amp_hi_log2"], sig["noise_sigma"] = 2.0, 5.0, 2.0   # very quiet: zero runs everywhere
        try:
            b = synth.make_batch(d, sig, want_pcm=True)
        except RuntimeError:      # the synthetic encoder refuses a few random recipes (a value it cannot represent)
            skipped += 1
            continue
        kb = int(rng.choice([14, 14, 14, 9, 16, 17, 24, 33, 40, 255]))      # rice_kmodifier: any byte but 0 (AlacFile.cs:82)
        d["rice_kmodifier"] = kb
        try:
            b = synth.make_batch(d, sig, want_pcm=True) if kb != 14 else b
        except RuntimeError:
            skipped += 1
            continue
        cfgs = [(4096, 24 if is24 else 16, 40, 10, kb, 2 if stereo else 1)]
        nc = 2 if stereo else 1
        b["slot_ints"] = int(d["n"].max()) * nc          # (one-channel elements in a two-channel stream still fill two channels)
        o = orc.decode_batch(orc.make_cfgs(cfgs), b["blob"], b["offsets"], b["sizes"], None, b["slot_ints"], n_threads=8)
        assert (o[3] == 0).all(), o[3]
        for variant in ("8-packet-16step", "dense", "8-packet-96reg", "8-packet"):   # the builds of the main kernel (ALACGPU_DENSE is read at create time)
            os.environ["ALACGPU_DENSE"] = {"8-packet-16step": "3", "dense": "1", "8-packet-96reg": "2", "8-packet": "4"}[variant]
            with pkg.AlacGpuContext(cfgs) as ctx:
                g = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], None, b["slot_ints"])
                assert np.array_equal(g[3], o[3]) and np.array_equal(g[1], o[1]) and np.array_equal(g[2], o[2]), (rounds, variant)
                for p in range(count):
                    cnt = int(d["n"][p]) * nc
                    if not np.array_equal(g[0][p, :cnt], o[0][p, :cnt]):
                        bad = np.nonzero(g[0][p, :cnt] != o[0][p, :cnt])[0]
                        raise SystemExit(f"MISMATCH round {rounds} seed {seed} variant {variant} packet {p} order {d['pred_order'][p]} "
                                         f"n {d['n'][p]} first bad index {bad[:5]}")
                if rounds % 4 == 0:      # the packed little-endian format (FormatSamples fused into the store)
                    ctx.set_output_format(1)
                    gp = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], None, b["slot_ints"])
                    bps = 3 if is24 else 2
                    for p in range(count):
                        cnt = int(d["n"][p]) * nc
                        v = o[0][p, :cnt].astype(np.int64)
                        exp = np.stack([(v >> (8 * k)) & 0xFF for k in range(bps)], axis=1).astype(np.uint8).reshape(-1)
                        got = gp[0][p].view(np.uint8)[: cnt * bps]
                        if not np.array_equal(got, exp):
                            raise SystemExit(f"PACKED MISMATCH round {rounds} seed {seed} variant {variant} packet {p}")
        rounds += 1
        packets += count
        if time.time() - last_note > 30:    # a long run has to show signs of life (gpurun kills silent commands)
            last_note = time.time()
            print(f"... {rounds} rounds, {packets} packets, {time.time() - t0:.0f} s", flush=True)
    print(f"stress ok: {rounds} rounds, {packets} packets x 4 builds of the main kernel, {time.time() - t0:.0f} s, seed {seed}, {skipped} recipes skipped")


if __name__ == "__main__":
    main()

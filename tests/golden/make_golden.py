#!/usr/bin/env python3
"""Generates tests/golden/golden_v2.npz: input packets and the outputs the LITERAL restatement of AlacFile.cs
(oracle/alacfile_literal.py) gives for them.

The reference (teekay/ALAC.NET) holds no fixtures and cannot run here (C#, no .NET) -- parity stays "unpinned" by
reference-held data.  What this file adds over round 1's golden_v1 (whose expected outputs came from the C oracle, i.e.
from the thing under test): the expected outputs now come from a second, independent, statement-by-statement reading of
the reference that shares no code with the C oracle, the synthetic encoder or the kernels.  Packets: the synthetic
encoder's (assorted orders incl. 0 / 17..30 / 31, quantisers incl. 0, Rice modifiers, mix weights incl. > 127, shift
bytes, mono / stereo, escapes incl. 24-bit, short frames, loud and silent content) plus hand-packed edge cases.
Build-container only (the pure-Python decoder takes a minute):  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

from alac.net_amd import synth  # noqa: E402
import alacfile_literal as lit  # noqa: E402
from bitpack import pack  # noqa: E402

EXC_NONE, EXC_SAMPLE_SIZE, EXC_PREDTYPE, EXC_INDEX, EXC_ARRAYCOPY = 0, 2, 3, 5, 6


def classify(e):
    if isinstance(e, IndexError):
        return EXC_INDEX
    if isinstance(e, ValueError):
        return EXC_ARRAYCOPY
    if "unimplemented sample size" in str(e):
        return EXC_SAMPLE_SIZE
    if "unhandled predicition type" in str(e):
        return EXC_PREDTYPE
    raise e


def main():
    N = 40
    d = synth.packet_descs(N, n=256, max_samples_per_frame=256)
    cfgs = [(256, 16, 40, 10, 14, 2), (256, 24, 40, 10, 14, 2), (256, 16, 40, 10, 14, 1), (256, 24, 40, 10, 14, 1),
            (256, 16, 255, 3, 9, 2), (256, 16, 40, 10, 14, 2)]
    ci = np.zeros(N, dtype=np.uint16)
    # 0-5: 16-bit stereo, assorted orders / mixes
    d["pred_order"][0:6, 0] = [8, 0, 31, 4, 17, 30]
    d["pred_order"][0:6, 1] = [8, 31, 0, 16, 30, 5]
    d["mix_weight"][0:6] = [1, 0, 1, 2, 0, 1]
    d["mix_shift"][0:6] = [2, 0, 1, 2, 0, 3]
    # 6-11: 24-bit stereo with ub 0/1/2
    d["sample_size"][6:12] = 24
    d["ub"][6:12] = [0, 1, 2, 0, 1, 2]
    d["pred_order"][6:12] = 16
    d["pred_order"][9:12, 0] = [20, 8, 31]
    ci[6:12] = 1
    # 12-15: mono 16 / 24
    d["stereo"][12:16] = 0
    d["sample_size"][14:16] = 24
    d["ub"][15] = 1
    ci[12:14] = 2
    ci[14:16] = 3
    # 16-17: escape packets (16-bit, 24-bit stereo: the Readbits(16) << 8 | Readbits(8) path); 18-19: short, hassize
    d["escape"][16:18] = 1
    d["sample_size"][17] = 24
    ci[17] = 1
    d["n"][18:20] = [1, 77]
    # 20: other quantisation / rice modifier; 21: pred type in a stereo element (throws); 22: unknown element
    d["quant"][20] = [4, 12]
    d["ricemod"][20] = [2, 7]
    d["pred_type"][21] = [1, 0]
    d["channels_field"][22] = 5
    # 24: interlacingLeftweight read unsigned, > 127 (App. B Q3); 25: weight 255 with a big shift
    d["mix_weight"][24], d["mix_shift"][24] = 200, 8
    d["mix_weight"][25], d["mix_shift"][25] = 255, 2
    # 26: quantiser 0 -> 1 << (q - 1) wraps to int.MinValue (Q12); 27: quantiser 15
    d["quant"][26] = [0, 0]
    d["quant"][27] = [15, 1]
    # 28-29: orders 20 / 23 / 29 / 18 with random coefficients (both adaptation signs over many taps)
    d["pred_order"][28] = [20, 23]
    d["pred_order"][29] = [29, 18]
    d["coef_mode"][28:30] = 1
    d["coefs"][28:30] = np.random.default_rng(28).integers(-2000, 2000, (2, 2, 32))
    # 30: mono element inside a two-channel stream (Q7/Q22); 31: mono escape 24-bit
    d["stereo"][30] = 0
    d["stereo"][31], d["sample_size"][31], d["escape"][31] = 0, 24, 1
    ci[31] = 3
    # 32: mono, unknown prediction type: the reference does NOT throw here (AlacFile.cs:484-496)
    d["stereo"][32] = 0
    d["pred_type"][32] = [2, 0]
    ci[32] = 2
    # 33: other Rice parameters (history mult 255, initial history 3, k modifier 9)
    d["rice_history_mult"][33], d["rice_initial_history"][33], d["rice_kmodifier"][33] = 255, 3, 9
    ci[33] = 4
    # 34: rice modifier 0 -> history never moves, zero-run symbol after every value; first run sees history 0 -> k 16 (Q1)
    d["ricemod"][34] = [0, 0]
    d["n"][34] = 64
    # 35-39: content that lives on the slow paths
    sig = synth.default_signal(0x601D)
    sig["silence_prob"] = 0.5
    sig["silence_min"], sig["silence_max"] = 32, 200
    b = synth.make_batch(d[:35], sig, want_pcm=False, n_threads=1)
    loud = synth.default_signal(0x10AD)
    loud["amp_lo_log2"], loud["amp_hi_log2"], loud["noise_sigma"] = 14.5, 15.0, 12000.0    # escapes, history saturation (Q19)
    quiet = synth.default_signal(0x5117)
    quiet["amp_lo_log2"], quiet["amp_hi_log2"], quiet["noise_sigma"] = 1.0, 3.0, 1.0         # zero runs everywhere (Q20)
    quiet["silence_prob"] = 1.0
    quiet["silence_min"], quiet["silence_max"] = 1, 40
    d["sample_size"][37] = 24
    ci[37] = 1
    d["stereo"][38] = 0
    ci[38] = 2
    b2 = synth.make_batch(d[35:37], loud, want_pcm=False, n_threads=1)
    b3 = synth.make_batch(d[37:40], quiet, want_pcm=False, n_threads=1)
    packets = []
    for bb in (b, b2, b3):
        for o, s in zip(bb["offsets"], bb["sizes"]):
            packets.append(bytes(bb["blob"][int(o):int(o) + int(s)]))
    # 23: a copy of packet 0 cut to a third: the reference reads on into its zero-filled 80 KiB buffer (Q8)
    packets[23] = packets[0][: len(packets[0]) // 3]
    cfg_rows = [int(c) for c in ci]
    # ---- hand-packed edge cases ----
    hand = [
        # zero-run symbol with history 0 and k = 16 whose run would leave the 16384-entry scratch (IndexOutOfRange, :242):
        # mono n=3, ricemod 0: value '0' -> dv 0, history stays 10 -> run k=4: '0'+'0001' = e 1 -> block 0 (un-read);
        # value '0' -> history 0 ... run k=16: nine 1s -> raw 16 bits 0xFFFF -> block 65535 -> throws
        (2, pack([(3, 0), (4, 0), (12, 0), (1, 1), (2, 0), (1, 0), (32, 3), (8, 0), (8, 0), (4, 0), (4, 0), (3, 0), (5, 0),
                  "0", "0", "000", "0", (9, 0x1FF), (16, 0xFFFF)], slack=4)),
        # hassize sample count above the scratch size (IndexOutOfRange in the store loop)
        (0, pack([(3, 1), (4, 0), (12, 0), (1, 1), (2, 0), (1, 1), (32, 20000), (16, 1), (16, 2)], slack=4)),
        # order 0 with more than 4096 samples would need Array.Copy past the array (Q2); frame length is 256 here, so
        # use a hassize count of 5000 on an escape-free mono packet of zeros: '0' bits all the way (value 0 each ...
        # the history decays into zero runs); only the exception class matters
        (2, pack([(3, 0), (4, 0), (12, 0), (1, 1), (2, 0), (1, 0), (32, 5000), (8, 0), (8, 0), (4, 0), (4, 9), (3, 4), (5, 0)], slack=2000)),
        # 16-bit stereo, shift bytes present (ub = 1): Deinterlace16 ignores them after reading them (:634-641, :705)
        (0, pack([(3, 1), (4, 0), (12, 0), (1, 1), (2, 1), (1, 0), (32, 2), (8, 0), (8, 0), (4, 0), (4, 0), (3, 4), (5, 0),
                  (4, 0), (4, 0), (3, 4), (5, 0), (8, 0x11), (8, 0x22), (8, 0x33), (8, 0x44), "110", "0", "10", "0", "0", "0"], slack=4)),
        # a two-channel element in a one-channel stream (cfg 2): out[i] = left, out[i + 1] = right (:353-354) -> the left channel
        (2, pack([(3, 1), (4, 0), (12, 0), (1, 1), (2, 0), (1, 0), (32, 1), (8, 8), (8, 200), (4, 0), (4, 0), (3, 4), (5, 0),
                  (4, 0), (4, 0), (3, 4), (5, 0), (9, 0x1FF), (17, 200), (9, 0x1FF), (17, 20)], slack=4)),
        (2, pack([(3, 1), (4, 0), (12, 0), (1, 1), (2, 0), (1, 1), (32, 3), (16, 5), (16, 6), (16, 0xFFF0), (16, 8), (16, 9), (16, 10)], slack=4)),
    ]
    for c, pk in hand:
        packets.append(pk)
        cfg_rows.append(c)
    n_all = len(packets)
    slot = 5120   # (room for the 5000-sample hand packet)
    pcm = np.zeros((n_all, slot), dtype=np.int32)
    ref_ret = np.full(n_all, -1, dtype=np.int64)
    ref_exc = np.zeros(n_all, dtype=np.int32)
    n_samples = np.zeros(n_all, dtype=np.int32)
    for p, pk in enumerate(packets):
        cfg = cfgs[cfg_rows[p]]
        try:
            out, ret = lit.decode_packet(cfg, pk)
        except Exception as e:   # noqa: BLE001 -- the reference's own exception types, classified
            ref_exc[p] = classify(e)
            continue
        ref_ret[p] = ret
        bytespersample = (cfg[1] // 8) * cfg[5]
        n = ret // bytespersample
        n_samples[p] = n
        if 0 < n * cfg[5] <= slot:
            pcm[p, : n * cfg[5]] = lit.canonical_from_reference_layout(out, n, cfg[1], cfg[5])
        print(f"packet {p}: {len(pk)} bytes, cfg {cfg}, return {ret}", flush=True)
    sizes = np.array([len(x) for x in packets], dtype=np.uint32)
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    blob = np.frombuffer(b"".join(packets) + bytes(16), dtype=np.uint8)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v2.npz")
    np.savez_compressed(out, blob=blob, offsets=offsets, sizes=sizes, cfg_idx=np.array(cfg_rows, dtype=np.uint16),
                        cfgs=np.array(cfgs, dtype=np.int64), pcm=pcm, ref_ret=ref_ret, ref_exc=ref_exc, n_samples=n_samples,
                        slot_ints=np.int64(slot))
    print(out, os.path.getsize(out), "bytes; exceptions", ref_exc.tolist())


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.npz: small input/output vectors for the decode path.

The reference (teekay/ALAC.NET) holds no fixtures and cannot run here (C#, no .NET), so these
vectors are made by this repo's own tools: packets by the synthetic encoder (alac.net_amd/synth),
expected outputs by the CPU oracle (oracle/alac_oracle.c, itself pinned by the hand-derived KATs in
tests/test_oracle_kat.py).  They freeze today's behaviour so that neither the oracle nor the GPU path
can drift silently.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]

from alac.net_amd import synth  # noqa: E402
import alac_oracle_py as orc  # noqa: E402


def main():
    N = 24
    d = synth.packet_descs(N, n=256, max_samples_per_frame=256)
    cfgs = [(256, 16, 40, 10, 14, 2), (256, 24, 40, 10, 14, 2), (256, 16, 40, 10, 14, 1), (256, 24, 40, 10, 14, 1)]
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
    # 16-17: escape packets; 18-19: short packets with hassize
    d["escape"][16:18] = 1
    d["sample_size"][17] = 24
    ci[17] = 1
    d["n"][18:20] = [1, 77]
    # 20: different quantisation / rice modifier; 21: pred type (status 3); 22: bad element (status 1)
    d["quant"][20] = [4, 12]
    d["ricemod"][20] = [2, 7]
    d["pred_type"][21] = [1, 0]
    d["channels_field"][22] = 5
    sig = synth.default_signal(0x601D)
    sig["silence_prob"] = 0.5
    sig["silence_min"], sig["silence_max"] = 32, 200
    b = synth.make_batch(d, sig, want_pcm=False, n_threads=1)
    # 23: truncated copy of packet 0 (status 5)
    b["sizes"][23] = b["sizes"][23] // 3
    slot = 512
    pcm, ob, os_, st = orc.decode_batch(orc.make_cfgs(cfgs), b["blob"], b["offsets"], b["sizes"], ci, slot)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v1.npz")
    np.savez_compressed(out, blob=b["blob"], offsets=b["offsets"], sizes=b["sizes"], cfg_idx=ci,
                        cfgs=np.array(cfgs, dtype=np.int64), pcm=pcm, out_bytes=ob, out_samples=os_, status=st,
                        slot_ints=np.int64(slot))
    print(out, os.path.getsize(out), "bytes; statuses", st.tolist())


if __name__ == "__main__":
    main()

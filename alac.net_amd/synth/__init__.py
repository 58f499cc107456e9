"""Synthetic ALAC packet generator (ctypes binding of synth/alac_synth.c).

The reference ships no audio, encoder or fixtures, so tests and bench.py
make their packets here.  `make_config_batch` builds the concrete inputs BASELINE.json's
configs name.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PKT_DTYPE = np.dtype(
    [
        ("n", "<u4"),
        ("max_samples_per_frame", "<u4"),
        ("sample_size", "u1"),
        ("stereo", "u1"),
        ("ub", "u1"),
        ("escape", "u1"),
        ("force_hassize", "u1"),
        ("pred_order", "u1", (2,)),
        ("quant", "u1", (2,)),
        ("ricemod", "u1", (2,)),
        ("pred_type", "u1", (2,)),
        ("mix_shift", "u1"),
        ("mix_weight", "u1"),
        ("coef_mode", "u1"),
        ("rice_history_mult", "u1"),
        ("rice_initial_history", "u1"),
        ("rice_kmodifier", "u1"),
        ("channels_field", "i1"),
        ("pad", "u1", (4,)),
        ("coefs", "<i2", (2, 32)),
    ]
)
assert PKT_DTYPE.itemsize == 160, PKT_DTYPE.itemsize

SIG_DTYPE = np.dtype(
    [
        ("seed", "<u8"),
        ("amp_lo_log2", "<f4"),
        ("amp_hi_log2", "<f4"),
        ("noise_sigma", "<f4"),
        ("silence_prob", "<f4"),
        ("silence_min", "<u4"),
        ("silence_max", "<u4"),
        ("lr_corr", "<f4"),
        ("_pad", "<u4"),
    ]
)
assert SIG_DTYPE.itemsize == 40


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libalacsynth.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.alac_synth_encode_packet.restype = C.c_size_t
        L.alac_synth_encode_packet.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.alac_synth_make_pcm.restype = None
        L.alac_synth_make_pcm.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.c_void_p]
        L.alac_synth_make_batch.restype = C.c_size_t
        L.alac_synth_make_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p,
                                            C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.alac_synth_max_packet_bytes.restype = C.c_size_t
        L.alac_synth_max_packet_bytes.argtypes = [C.c_uint32, C.c_int, C.c_int]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def default_signal(seed):
    s = np.zeros(1, dtype=SIG_DTYPE)
    s["seed"] = seed
    s["amp_lo_log2"], s["amp_hi_log2"] = 12.0, 14.0
    s["noise_sigma"] = 64.0
    s["silence_prob"] = 0.05
    s["silence_min"], s["silence_max"] = 256, 2048
    s["lr_corr"] = 0.8
    return s


def packet_descs(n_packets, **kw):
    """Array of packet recipes with the stream defaults of the reference's comments
    (AlacFile.cs:43-45: historyMult 40, initialHistory 10, kModifier 14) and SURVEY 8(d)'s
    per-channel header (predictionType 0, q 9, ricemodifier 4)."""
    d = np.zeros(n_packets, dtype=PKT_DTYPE)
    d["n"] = 4096
    d["max_samples_per_frame"] = 4096
    d["sample_size"] = 16
    d["stereo"] = 1
    d["pred_order"] = 8
    d["quant"] = 9
    d["ricemod"] = 4
    d["mix_shift"] = 2
    d["mix_weight"] = 1
    d["rice_history_mult"], d["rice_initial_history"], d["rice_kmodifier"] = 40, 10, 14
    d["channels_field"] = -1
    for k, v in kw.items():
        d[k] = v
    return d


def encode_packet(desc, pcm):
    """Encode one packet from interleaved int32 PCM; returns bytes."""
    desc = np.ascontiguousarray(desc).reshape(1)
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    cap = int(lib().alac_synth_max_packet_bytes(int(desc["n"][0]), int(desc["sample_size"][0]), int(desc["stereo"][0])))
    out = np.zeros(cap, dtype=np.uint8)
    sz = lib().alac_synth_encode_packet(_ptr(desc), _ptr(pcm), _ptr(out), cap)
    if sz == 0:
        raise ValueError("alac_synth_encode_packet failed (bad recipe?)")
    return out[:sz].tobytes()


def make_pcm(sig, index, sample_size, ch, n):
    pcm = np.zeros(n * ch, dtype=np.int32)
    lib().alac_synth_make_pcm(_ptr(sig), index, sample_size, ch, n, _ptr(pcm))
    return pcm


def make_batch(descs, sig, first_index=0, n_threads=None, want_pcm=False):
    """Generate + encode a batch.  Returns dict(blob, offsets, sizes, pcm|None, slot_ints)."""
    descs = np.ascontiguousarray(descs)
    n = len(descs)
    if n_threads is None:
        n_threads = max(1, min(16, len(os.sched_getaffinity(0))))
    # realistic packets are well under half the worst case; grow on demand
    est = sum(int(x) for x in (descs["n"].astype(np.int64) * (1 + descs["stereo"].astype(np.int64)) * 4)) + 64 * n + 4096
    offsets = np.zeros(n, dtype=np.uint64)
    sizes = np.zeros(n, dtype=np.uint32)
    slot = int((descs["n"].astype(np.int64) * (1 + descs["stereo"].astype(np.int64))).max()) if n else 0
    pcm = np.zeros((n, slot), dtype=np.int32) if want_pcm else None
    for attempt in range(3):
        blob = np.zeros(est, dtype=np.uint8)
        used = lib().alac_synth_make_batch(_ptr(descs), n, _ptr(sig), first_index, n_threads, _ptr(blob), est,
                                           _ptr(offsets), _ptr(sizes), _ptr(pcm), slot)
        if used:
            # keep >= 16 bytes of slack after the last packet so vector loads near the end stay in-bounds
            return dict(blob=blob[: used + 16].copy(), offsets=offsets, sizes=sizes, pcm=pcm, slot_ints=slot)
        est *= 4
    raise RuntimeError("alac_synth_make_batch failed")


# ---- BASELINE.json configs as concrete inputs -------------------------
def config_descs(cfg, n_packets=None, seed=None):
    """Returns (descs, signal, stream_cfgs, cfg_idx) for BASELINE config `cfg` in {1,2,3,4,5}.
    stream_cfgs: list of (max_samples_per_frame, sample_size, pb, mb, kb, num_channels)."""
    defaults = {1: 2584, 2: 4096, 3: 8192, 4: 65536, 5: 32768}
    n = n_packets if n_packets is not None else defaults[cfg]
    seed = (0xA1AC0000 + (cfg << 24)) if seed is None else seed
    sig = default_signal(seed)
    idx = np.arange(n)
    if cfg in (1, 2):
        d = packet_descs(n)
        d["mix_weight"] = np.where(idx % 2 == 0, 1, 0)
        return d, sig, [(4096, 16, 40, 10, 14, 2)], None
    if cfg == 3:
        d = packet_descs(n, n=8192, max_samples_per_frame=8192, sample_size=24, pred_order=16)
        d["ub"] = idx % 2
        d["mix_weight"] = np.where(idx % 2 == 0, 1, 0)
        return d, sig, [(8192, 24, 40, 10, 14, 2)], None
    if cfg == 4:
        d = packet_descs(n, stereo=0, mix_shift=0, mix_weight=0)
        return d, sig, [(4096, 16, 40, 10, 14, 1)], None
    if cfg == 5:
        rng = np.random.default_rng(seed)
        d = packet_descs(n)
        is24 = rng.integers(0, 2, n).astype(bool)
        d["sample_size"] = np.where(is24, 24, 16)
        orders = np.concatenate([np.arange(4, 31), [31]])
        d["pred_order"][:, 0] = orders[rng.integers(0, len(orders), n)]
        d["pred_order"][:, 1] = orders[rng.integers(0, len(orders), n)]
        d["ub"] = np.where(is24, rng.integers(0, 2, n), 0)
        d["mix_weight"] = rng.integers(0, 2, n)
        d["escape"] = rng.random(n) < 0.02
        short = rng.random(n) < 0.02
        d["n"] = np.where(short, rng.integers(1, 4096, n), 4096)
        return d, sig, [(4096, 16, 40, 10, 14, 2), (4096, 24, 40, 10, 14, 2)], is24.astype(np.uint16)
    raise ValueError(cfg)


def make_config_batch(cfg, n_packets=None, seed=None, n_threads=None, want_pcm=False, first_index=0):
    d, sig, stream_cfgs, cfg_idx = config_descs(cfg, n_packets, seed)
    b = make_batch(d, sig, first_index=first_index, n_threads=n_threads, want_pcm=want_pcm)
    b.update(descs=d, stream_cfgs=stream_cfgs, cfg_idx=cfg_idx)
    return b

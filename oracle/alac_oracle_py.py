"""ctypes binding of the CPU ORACLE (oracle/alac_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker / the reported CPU baseline -- never as the product path.  The product
(alac.net_amd) must not import anything from oracle/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OracleCfg(C.Structure):
    _fields_ = [
        ("max_samples_per_frame", C.c_uint32),
        ("sample_size", C.c_uint8),
        ("rice_history_mult", C.c_uint8),
        ("rice_initial_history", C.c_uint8),
        ("rice_kmodifier", C.c_uint8),
        ("num_channels", C.c_uint8),
        ("ctor_sample_size", C.c_uint8),
        ("reserved", C.c_uint8),
    ]


CFG_DTYPE = np.dtype(
    [
        ("max_samples_per_frame", "<u4"),
        ("sample_size", "u1"),
        ("rice_history_mult", "u1"),
        ("rice_initial_history", "u1"),
        ("rice_kmodifier", "u1"),
        ("num_channels", "u1"),
        ("ctor_sample_size", "u1"),
        ("reserved", "u1"),
        ("_pad", "u1"),
    ]
)
assert CFG_DTYPE.itemsize == C.sizeof(OracleCfg) == 12

ST_OK, ST_UNSUPPORTED_ELEMENT, ST_UNSUPPORTED_SAMPLE_SIZE, ST_UNSUPPORTED_PREDTYPE = 0, 1, 2, 3
ST_BAD_SAMPLE_COUNT, ST_OVERRUN, ST_REF_THROWS, ST_UNSUPPORTED_PARAMS = 4, 5, 6, 7


def build(asan=False):
    target = "libalac_oracle_asan.so" if asan else "libalac_oracle.so"
    subprocess.run(["make", "-C", _HERE, target], check=True, capture_output=True)
    return os.path.join(_HERE, target)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libalac_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.alac_oracle_decode_frame.restype = C.c_int
        L.alac_oracle_decode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                               C.c_void_p, C.c_void_p]
        L.alac_oracle_decode_batch.restype = C.c_int
        L.alac_oracle_decode_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_int]
        L.alac_oracle_expand_reference_layout.restype = C.c_size_t
        L.alac_oracle_expand_reference_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.alac_oracle_format_samples.restype = C.c_size_t
        L.alac_oracle_format_samples.argtypes = [C.c_int, C.c_void_p, C.c_int32, C.c_void_p]
        L.alac_oracle_count_leading_zeros.restype = C.c_int
        L.alac_oracle_count_leading_zeros.argtypes = [C.c_int32]
        L.alac_oracle_predictor.restype = None
        L.alac_oracle_predictor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.alac_oracle_deinterlace16.restype = None
        L.alac_oracle_deinterlace16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                C.c_int]
        L.alac_oracle_rice_decode.restype = C.c_int
        L.alac_oracle_rice_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_void_p]
        L.alac_oracle_set_info.restype = None
        L.alac_oracle_set_info.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


def make_cfgs(rows):
    """rows: iterable of dicts/tuples (max_samples_per_frame, sample_size, pb, mb, kb, num_channels)."""
    arr = np.zeros(len(rows), dtype=CFG_DTYPE)
    for i, r in enumerate(rows):
        if isinstance(r, dict):
            for k, v in r.items():
                arr[i][k] = v
        else:
            (arr[i]["max_samples_per_frame"], arr[i]["sample_size"], arr[i]["rice_history_mult"],
             arr[i]["rice_initial_history"], arr[i]["rice_kmodifier"], arr[i]["num_channels"]) = r
    return arr


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def decode_frame(cfg_row, packet, capacity=None):
    """Decode one packet.  Returns (status, pcm[int32 n*nc], out_bytes, out_samples)."""
    cfgs = make_cfgs([cfg_row]) if not isinstance(cfg_row, np.ndarray) else cfg_row
    pkt = np.frombuffer(bytes(packet), dtype=np.uint8)
    nc = int(cfgs[0]["num_channels"])
    cap = capacity or (16384 * max(nc, 1) + 8)
    pcm = np.zeros(cap, dtype=np.int32)
    ob = C.c_int32(0)
    os_ = C.c_int32(0)
    st = lib().alac_oracle_decode_frame(_ptr(cfgs), _ptr(pkt) if len(pkt) else None, len(pkt), _ptr(pcm), cap,
                                        C.byref(ob), C.byref(os_))
    n = os_.value
    cnt = n * nc if (st == ST_OK and 0 < n <= 16384) else 0
    return st, pcm[:cnt].copy(), ob.value, n


def decode_batch(cfgs, blob, offsets, sizes, cfg_idx, slot_ints, n_threads=1):
    """Same argument meaning as alacgpu_decode_batch.  Returns (pcm[n, slot], out_bytes, out_samples, status)."""
    n = len(sizes)
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
    ci = None if cfg_idx is None else np.ascontiguousarray(cfg_idx, dtype=np.uint16)
    pcm = np.zeros((n, slot_ints), dtype=np.int32)
    ob = np.zeros(n, dtype=np.int32)
    os_ = np.zeros(n, dtype=np.int32)
    st = np.zeros(n, dtype=np.int32)
    rc = lib().alac_oracle_decode_batch(_ptr(cfgs), len(cfgs), _ptr(blob), _ptr(offsets), _ptr(sizes), _ptr(ci), n,
                                        _ptr(pcm), slot_ints, _ptr(ob), _ptr(os_), _ptr(st), n_threads)
    if rc != 0:
        raise RuntimeError("oracle batch failed")
    return pcm, ob, os_, st


def expand_reference_layout(cfg_row, pcm, n_samples):
    cfgs = make_cfgs([cfg_row]) if not isinstance(cfg_row, np.ndarray) else cfg_row
    nc = int(cfgs[0]["num_channels"])
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    out = np.zeros(n_samples * nc * 3 + 8, dtype=np.int32)
    cnt = lib().alac_oracle_expand_reference_layout(_ptr(cfgs), _ptr(pcm), n_samples, _ptr(out))
    return out[:cnt].copy()


def format_samples(bps, ref_ints, count_bytes):
    ref_ints = np.ascontiguousarray(ref_ints, dtype=np.int32)
    dst = np.zeros(max(count_bytes, 0) + 8, dtype=np.uint8)
    cnt = lib().alac_oracle_format_samples(bps, _ptr(ref_ints), count_bytes, _ptr(dst))
    return dst[:cnt].copy()


def clz(x):
    return lib().alac_oracle_count_leading_zeros(C.c_int32(x))


def predictor(err, rss, coef, q):
    buf = np.array(err, dtype=np.int32)
    cf = np.zeros(32, dtype=np.int32)
    cf[: len(coef)] = coef
    lib().alac_oracle_predictor(_ptr(buf), len(buf), rss, _ptr(cf), len(coef), q)
    return buf, cf[: len(coef)].copy()


def deinterlace16(a, b, nc, shift, weight):
    a = np.array(a, dtype=np.int32)
    b = np.array(b, dtype=np.int32)
    out = np.zeros(len(a) * nc + 1, dtype=np.int32)
    lib().alac_oracle_deinterlace16(_ptr(a), _ptr(b), _ptr(out), nc, len(a), shift, weight)
    return out[: len(a) * nc]


def rice_decode(bits, n, rss, init_hist, kmod, hist_mult):
    bits = np.frombuffer(bytes(bits), dtype=np.uint8)
    out = np.zeros(n, dtype=np.int32)
    end = C.c_int(0)
    st = lib().alac_oracle_rice_decode(_ptr(bits), len(bits), _ptr(out), n, rss, init_hist, kmod, hist_mult,
                                       C.byref(end))
    return st, out, end.value


def set_info(codec_data_ints, samplesize, numchannels):
    arr = np.ascontiguousarray(codec_data_ints, dtype=np.int32)
    cfg = np.zeros(1, dtype=CFG_DTYPE)
    lib().alac_oracle_set_info(_ptr(arr), samplesize, numchannels, _ptr(cfg))
    return cfg

"""alac.net_amd -- MI355X-native ALAC frame-decode path (teekay/ALAC.NET's AlacFile.DecodeFrame).

Python host side over the C ABI of include/alacgpu.h (libalacgpu.so: hand-written gfx950 HIP
kernels).  It mirrors the reference's interface for the path -- `AlacFile(samplesize,
numchannels)`, `SetInfo(codecData)`, `DecodeFrame(inbuffer, outbuffer)` (AlacFile.cs:16,:63,:428)
-- and adds the batch entry points.  There is NO CPU fallback: if libalacgpu.so is missing or no
gfx950 GPU is usable, every entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("ALACGPU_LIB", os.path.join(_HERE, "csrc", "libalacgpu.so"))  # override: A/B builds only
_LIB = None

# per-packet status codes (include/alacgpu.h)
ST_OK, ST_UNSUPPORTED_ELEMENT, ST_UNSUPPORTED_SAMPLE_SIZE, ST_UNSUPPORTED_PREDTYPE = 0, 1, 2, 3
ST_BAD_SAMPLE_COUNT, ST_OVERRUN, ST_REF_THROWS, ST_UNSUPPORTED_PARAMS = 4, 5, 6, 7

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
assert CFG_DTYPE.itemsize == 12

# every symbol include/alacgpu.h declares: (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "alacgpu_version": (C.c_int, []),
    "alacgpu_device_count": (C.c_int, []),
    "alacgpu_alloc_pinned": (_VP, [C.c_size_t]),
    "alacgpu_free_pinned": (None, [_VP]),
    "alacgpu_create": (C.c_int, [_VP, C.c_uint32, C.c_int, C.POINTER(_VP)]),
    "alacgpu_destroy": (None, [_VP]),
    "alacgpu_cfg_from_codec_data": (C.c_int, [_VP, C.c_uint32, C.c_int, C.c_int, _VP]),
    "alacgpu_decode_batch": (C.c_int, [_VP, _VP, C.c_uint64, _VP, _VP, _VP, C.c_uint32, _VP, C.c_uint32, _VP, _VP, _VP]),
    "alacgpu_decode_batch_sharded": (C.c_int, [_VP, C.c_uint32, _VP, C.c_uint64, _VP, _VP, _VP, C.c_uint32, _VP, C.c_uint32, _VP,
                                               _VP, _VP]),
    "alacgpu_decode_batch_device": (C.c_int, [_VP, _VP, C.c_uint64, _VP, _VP, _VP, C.c_uint32, _VP, C.c_uint32, _VP,
                                              _VP, _VP, _VP]),
    "alacgpu_decode_frame": (C.c_int, [_VP, C.c_uint32, _VP, C.c_uint32, _VP, C.c_uint32, _VP, _VP]),
    "alacgpu_expand_reference_layout": (C.c_size_t, [_VP, _VP, C.c_int32, _VP]),
    "alacgpu_format_samples": (C.c_size_t, [C.c_int, _VP, C.c_int32, _VP]),
    "alacgpu_last_kernel_ms": (C.c_float, [_VP]),
    "alacgpu_set_output_format": (C.c_int, [_VP, C.c_int]),
    "alacgpu_strerror": (C.c_char_p, [C.c_int]),
    "alacgpu_status_string": (C.c_char_p, [C.c_int]),
    "alacgpu_last_error": (C.c_char_p, [_VP]),
    "alacgpu_ctx_device": (C.c_int, [_VP]),
    "alacgpu_shard_ranges": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _VP]),
    "alacgpu_comm_get_unique_id": (C.c_int, [_VP]),
    "alacgpu_comm_create": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.POINTER(_VP)]),
    "alacgpu_comm_destroy": (None, [_VP]),
    "alacgpu_comm_rank": (C.c_int, [_VP]),
    "alacgpu_comm_world": (C.c_int, [_VP]),
    "alacgpu_comm_last_error": (C.c_char_p, [_VP]),
    "alacgpu_allgather_pcm": (C.c_int, [_VP, _VP, _VP, C.c_uint32, _VP]),
    "alacgpu_decode_allgather_device": (C.c_int, [_VP, _VP, _VP, C.c_uint64, _VP, _VP, _VP, _VP, _VP, C.c_uint32, _VP, _VP, _VP,
                                                  C.c_uint32, _VP]),
}


class AlacGpuError(RuntimeError):
    pass


def lib():
    """Load libalacgpu.so (built by __graft_entry__.build()).  Fails loudly: no fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIBPATH):
            raise AlacGpuError(
                f"{_LIBPATH} is missing: build it with `make -C alac.net_amd/csrc` "
                "(or __graft_entry__.build()).  There is no CPU fallback."
            )
        L = C.CDLL(_LIBPATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(_VP) if a is not None else None


def make_cfgs(rows):
    """rows: iterable of (max_samples_per_frame, sample_size, pb, mb, kb, num_channels) tuples or dicts."""
    if isinstance(rows, np.ndarray) and rows.dtype == CFG_DTYPE:
        return np.ascontiguousarray(rows)
    arr = np.zeros(len(rows), dtype=CFG_DTYPE)
    for i, r in enumerate(rows):
        if isinstance(r, dict):
            for k, v in r.items():
                arr[i][k] = v
        else:
            (arr[i]["max_samples_per_frame"], arr[i]["sample_size"], arr[i]["rice_history_mult"],
             arr[i]["rice_initial_history"], arr[i]["rice_kmodifier"], arr[i]["num_channels"]) = r
    return arr


def _check(rc, ctx=None):
    if rc != 0:
        L = lib()
        msg = L.alacgpu_strerror(rc).decode()
        if ctx:
            detail = L.alacgpu_last_error(ctx).decode()
            if detail:
                msg += f" ({detail})"
        raise AlacGpuError(f"alacgpu rc={rc}: {msg}")


class AlacGpuContext:
    """Owns one alacgpu_ctx (one per host thread; calls are blocking unless stated)."""

    def __init__(self, cfgs, device=0):
        self._ctx = _VP()
        self.cfgs = make_cfgs(cfgs)
        L = lib()
        rc = L.alacgpu_create(_ptr(self.cfgs), len(self.cfgs), device, C.byref(self._ctx))
        if rc != 0:
            self._ctx = _VP()
        _check(rc)
        self.device = device

    def close(self):
        if self._ctx:
            lib().alacgpu_destroy(self._ctx)
            self._ctx = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- host buffers: H2D + kernel + D2H ---------------------------------------------------------
    def decode_batch(self, blob, offsets, sizes, cfg_idx=None, slot_ints=None, out=None):
        """Returns (pcm[n, slot_ints] int32, out_bytes[n], out_samples[n], status[n]).  `out`: a pcm array to decode
        into instead of a fresh one (a fresh 100+ MB array costs more in page faults than the whole decode)."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        ci = None if cfg_idx is None else np.ascontiguousarray(cfg_idx, dtype=np.uint16)
        n = len(sizes)
        if slot_ints is None:
            slot_ints = int(max(int(c["max_samples_per_frame"]) * int(c["num_channels"]) for c in self.cfgs))
        if out is not None:
            if out.dtype != np.int32 or out.shape != (n, slot_ints) or not out.flags.c_contiguous:
                raise ValueError("out must be a C-contiguous int32 array of shape (n_packets, slot_ints)")
            pcm = out
        else:
            pcm = np.zeros((n, slot_ints), dtype=np.int32)
        ob = np.zeros(n, dtype=np.int32)
        os_ = np.zeros(n, dtype=np.int32)
        st = np.zeros(n, dtype=np.int32)
        rc = lib().alacgpu_decode_batch(self._ctx, _ptr(blob), blob.size, _ptr(offsets), _ptr(sizes), _ptr(ci), n,
                                        _ptr(pcm), slot_ints, _ptr(ob), _ptr(os_), _ptr(st))
        _check(rc, self._ctx)
        return pcm, ob, os_, st

    # -- device buffers (torch tensors on this device), asynchronous on `stream` --------------------
    def decode_batch_device(self, d_blob, blob_bytes, d_offsets, d_sizes, d_cfg_idx, n_packets, d_pcm, slot_ints,
                            d_out_bytes, d_out_samples, d_status, stream=0):
        """All d_* are torch CUDA tensors (or None where the header allows NULL); `stream` is a raw
        hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream)."""
        def dp(t):
            return _VP(t.data_ptr()) if t is not None else None

        rc = lib().alacgpu_decode_batch_device(self._ctx, dp(d_blob), blob_bytes, dp(d_offsets), dp(d_sizes),
                                               dp(d_cfg_idx), n_packets, dp(d_pcm), slot_ints, dp(d_out_bytes),
                                               dp(d_out_samples), dp(d_status), _VP(stream))
        _check(rc, self._ctx)

    def set_output_format(self, fmt):
        """0: int32 per sample (default).  1: packed little-endian PCM bytes (FormatSamples fused into the store);
        packet p's bytes are pcm[p].view(uint8)[:out_bytes[p]]."""
        _check(lib().alacgpu_set_output_format(self._ctx, fmt), self._ctx)

    def last_kernel_ms(self):
        return float(lib().alacgpu_last_kernel_ms(self._ctx))

    def decode_frame(self, cfg_index, packet):
        """Single-packet DecodeFrame in the reference's own int[] layout.
        Returns (ref_ints, out_bytes, status)."""
        pkt = np.frombuffer(bytes(packet), dtype=np.uint8)
        cfg = self.cfgs[cfg_index]
        cap = 16384 * int(cfg["num_channels"]) * (3 if int(cfg["sample_size"]) == 24 else 1)
        out = np.zeros(cap, dtype=np.int32)
        ob = C.c_int32(0)
        st = C.c_int32(0)
        rc = lib().alacgpu_decode_frame(self._ctx, cfg_index, _ptr(pkt), len(pkt), _ptr(out), cap, C.byref(ob),
                                        C.byref(st))
        _check(rc, self._ctx)
        return out, ob.value, st.value


def shard_ranges(sizes, world):
    """alacgpu_shard_ranges: the packet partition of every multi-GPU entry point -- contiguous ranges cut at multiples of 8
    packets, balanced by packet bytes.  Returns first[world + 1]; rank r owns packets first[r] .. first[r+1].  Host arithmetic."""
    sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
    first = np.zeros(world + 1, dtype=np.uint32)
    _check(lib().alacgpu_shard_ranges(_ptr(sizes), len(sizes), world, _ptr(first)))
    return first


class AlacGpuComm:
    """alacgpu_comm: this rank's handle on the RCCL communicator for the all-gather of decoded PCM (one process per GPU).
    `unique_id()` on rank 0, hand the 128 bytes to the other ranks (torch.distributed broadcast, a file, MPI ...), then
    every rank constructs AlacGpuComm(ctx, id, rank, world) -- a collective call."""

    @staticmethod
    def unique_id():
        buf = np.zeros(128, dtype=np.uint8)
        rc = lib().alacgpu_comm_get_unique_id(_ptr(buf))
        if rc != 0:
            raise AlacGpuError(f"alacgpu rc={rc}: {lib().alacgpu_strerror(rc).decode()} ({lib().alacgpu_comm_last_error(None).decode()})")
        return buf

    def __init__(self, ctx, unique_id, rank, world):
        self._comm = _VP()
        self.ctx, self.rank, self.world = ctx, rank, world
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        assert uid.size == 128
        rc = lib().alacgpu_comm_create(ctx._ctx, _ptr(uid), rank, world, C.byref(self._comm))
        if rc != 0:
            self._comm = _VP()
            raise AlacGpuError(f"alacgpu rc={rc}: {lib().alacgpu_strerror(rc).decode()} ({lib().alacgpu_comm_last_error(None).decode()})")

    def _check(self, rc):
        if rc != 0:
            raise AlacGpuError(f"alacgpu rc={rc}: {lib().alacgpu_strerror(rc).decode()} ({lib().alacgpu_comm_last_error(self._comm).decode()})")

    def allgather_pcm(self, d_full, first, slot_ints, stream=0):
        """d_full: torch int32 CUDA tensor [n_packets, slot_ints] holding this rank's packets first[rank]..first[rank+1]
        decoded in place; asynchronous on `stream` (raw hipStream_t)."""
        first = np.ascontiguousarray(first, dtype=np.uint32)
        self._check(lib().alacgpu_allgather_pcm(self._comm, _VP(d_full.data_ptr()), _ptr(first), slot_ints, _VP(stream)))

    def decode_allgather_device(self, d_blob, blob_bytes, d_offsets, d_sizes, d_cfg_idx, first, d_full, slot_ints, d_out_bytes,
                                d_out_samples, d_status, n_chunks=4, stream=0):
        """Decode this rank's range (device arrays indexed by GLOBAL packet number) in n_chunks pieces and gather piece k
        while piece k+1 decodes; asynchronous on `stream`."""
        def dp(t):
            return _VP(t.data_ptr()) if t is not None else None
        first = np.ascontiguousarray(first, dtype=np.uint32)
        self._check(lib().alacgpu_decode_allgather_device(self.ctx._ctx, self._comm, dp(d_blob), blob_bytes, dp(d_offsets), dp(d_sizes),
                                                          dp(d_cfg_idx), _ptr(first), dp(d_full), slot_ints, dp(d_out_bytes),
                                                          dp(d_out_samples), dp(d_status), n_chunks, _VP(stream)))

    def close(self):
        if self._comm:
            lib().alacgpu_comm_destroy(self._comm)
            self._comm = _VP()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_batch_sharded(contexts, blob, offsets, sizes, cfg_idx=None, slot_ints=None, out=None):
    """alacgpu_decode_batch_sharded: one host batch over several AlacGpuContext objects (one per GPU) from this process.
    Returns (pcm[n, slot_ints] int32, out_bytes[n], out_samples[n], status[n]) like AlacGpuContext.decode_batch."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
    ci = None if cfg_idx is None else np.ascontiguousarray(cfg_idx, dtype=np.uint16)
    n = len(sizes)
    if slot_ints is None:
        slot_ints = int(max(int(c["max_samples_per_frame"]) * int(c["num_channels"]) for c in contexts[0].cfgs))
    pcm = out if out is not None else np.zeros((n, slot_ints), dtype=np.int32)
    if pcm.dtype != np.int32 or pcm.shape != (n, slot_ints) or not pcm.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous int32 array of shape (n_packets, slot_ints)")
    ob = np.zeros(n, dtype=np.int32)
    os_ = np.zeros(n, dtype=np.int32)
    st = np.zeros(n, dtype=np.int32)
    handles = (_VP * len(contexts))(*[c._ctx for c in contexts])
    rc = lib().alacgpu_decode_batch_sharded(handles, len(contexts), _ptr(blob), blob.size, _ptr(offsets), _ptr(sizes), _ptr(ci), n,
                                            _ptr(pcm), slot_ints, _ptr(ob), _ptr(os_), _ptr(st))
    _check(rc, contexts[0]._ctx)
    return pcm, ob, os_, st


class PinnedBuffer:
    """Page-locked host memory from alacgpu_alloc_pinned, viewed as a numpy array (batch buffers that are reused across
    calls: transfers from and to it run at link speed).  Free with close() / as a context manager."""

    def __init__(self, shape, dtype):
        self._shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self._dtype = np.dtype(dtype)
        nbytes = int(np.prod(self._shape, dtype=np.int64)) * self._dtype.itemsize
        self._p = lib().alacgpu_alloc_pinned(max(nbytes, 1))
        if not self._p:
            raise AlacGpuError("alacgpu_alloc_pinned failed")
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=self._dtype, count=int(np.prod(self._shape, dtype=np.int64))).reshape(self._shape)

    def close(self):
        if self._p:
            self.array = None
            lib().alacgpu_free_pinned(self._p)
            self._p = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    """Usable gfx950 devices (0: none -- creating a context then fails, there is no CPU fallback)."""
    return int(lib().alacgpu_device_count())


def expand_reference_layout(cfg_row, pcm, n_samples):
    cfgs = make_cfgs([cfg_row]) if not isinstance(cfg_row, np.ndarray) else cfg_row
    nc = int(cfgs[0]["num_channels"])
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    out = np.zeros(n_samples * nc * 3 + 8, dtype=np.int32)
    cnt = lib().alacgpu_expand_reference_layout(_ptr(cfgs), _ptr(pcm), n_samples, _ptr(out))
    return out[:cnt].copy()


def format_samples(bps, ref_ints, count_bytes):
    """AlacContext.FormatSamples (AlacContext.cs:214-256)."""
    ref_ints = np.ascontiguousarray(ref_ints, dtype=np.int32)
    dst = np.zeros(max(count_bytes, 0) + 8, dtype=np.uint8)
    cnt = lib().alacgpu_format_samples(bps, _ptr(ref_ints), count_bytes, _ptr(dst))
    return dst[:cnt].copy()


def cfg_from_codec_data(codec_data_ints, samplesize, numchannels):
    """AlacFile.SetInfo's parse (AlacFile.cs:63-93) of the int-per-byte CodecData array."""
    arr = np.ascontiguousarray(codec_data_ints, dtype=np.int32)
    cfg = np.zeros(1, dtype=CFG_DTYPE)
    _check(lib().alacgpu_cfg_from_codec_data(_ptr(arr), len(arr), samplesize, numchannels, _ptr(cfg)))
    return cfg


class AlacFile:
    """Mirror of the reference's `internal class AlacFile` surface for the path (AlacFile.cs:14-20,
    :63, :428) on the GPU library: same names, argument meaning and error behaviour; plus DecodeBatch."""

    def __init__(self, samplesize, numchannels, device=0):
        self._samplesize = samplesize
        self._numchannels = numchannels
        self._device = device
        self._ctx = None
        self._cfg = None

    def SetInfo(self, inputbuffer):
        self._cfg = cfg_from_codec_data(inputbuffer, self._samplesize, self._numchannels)
        if self._ctx is not None:
            self._ctx.close()
        self._ctx = AlacGpuContext(self._cfg, self._device)

    def _raise_for(self, st, predtype_hint=None):
        # the reference signals these by exceptions (AlacFile.cs:574,:650,:660,:715)
        if st == ST_UNSUPPORTED_SAMPLE_SIZE:
            raise Exception("FIXME: unimplemented sample size " + str(int(self._cfg[0]["sample_size"])))
        if st == ST_UNSUPPORTED_PREDTYPE:
            raise Exception("FIXME: unhandled predicition type")
        if st in (ST_BAD_SAMPLE_COUNT, ST_OVERRUN):
            raise IndexError("Index was outside the bounds of the array.")
        if st == ST_REF_THROWS:
            raise ValueError("Destination array was not long enough.")
        if st == ST_UNSUPPORTED_PARAMS:
            raise Exception("unsupported parameter combination")

    def DecodeFrame(self, inbuffer, outbuffer):
        """int DecodeFrame(byte[] inbuffer, int[] outbuffer): fills outbuffer in the reference's layout,
        returns the byte count (AlacFile.cs:718)."""
        if self._ctx is None:
            raise Exception("SetInfo must be called first")
        ref, out_bytes, st = self._ctx.decode_frame(0, inbuffer)
        if st == ST_UNSUPPORTED_ELEMENT:
            return out_bytes  # reference decodes nothing and still returns outputsize (:437,:577,:718)
        if (st == ST_UNSUPPORTED_SAMPLE_SIZE and len(inbuffer) and (int(inbuffer[0]) >> 5) == 1
                and int(self._cfg[0]["sample_size"]) not in (20, 32)):
            return out_bytes  # a two-channel element of any other sample size: nothing is written, no exception (:701-716)
        if st == ST_UNSUPPORTED_PREDTYPE and len(inbuffer) and (int(inbuffer[0]) >> 5) == 0:
            # one-channel element with a prediction type other than 0: the reference skips the predictor without a word and
            # hands out _outputsamplesBufferA (AlacFile.cs:484-496) -- which, once a compressed frame has been decoded, IS
            # the residual buffer (:486): the un-predicted residuals, which is what the library returns with status 3.
            # (A decoder that has never decoded a compressed frame would show zeros there: not reproduced)
            n = min(len(outbuffer), len(ref))
            outbuffer[:n] = ref[:n]
            return out_bytes
        self._raise_for(st)
        n = min(len(outbuffer), len(ref))
        outbuffer[:n] = ref[:n]
        return out_bytes

    def DecodeBatch(self, blob, offsets, sizes, slot_ints=None):
        """Batch-submit entry point (north_star: "AlacContext gains a batch-submit entry point")."""
        if self._ctx is None:
            raise Exception("SetInfo must be called first")
        return self._ctx.decode_batch(blob, offsets, sizes, None, slot_ints)

    def Dispose(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

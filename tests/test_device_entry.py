"""The entry point bench.py times -- alacgpu_decode_batch_device (device pointers, asynchronous on a HIP stream) -- with
torch tensors as the device memory: user streams, optional outputs, argument checks, several calls in flight on one ctx."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t

    assert t.cuda.device_count() > 0
    t.cuda.set_device(0)
    return t


@pytest.fixture(scope="module")
def pkg():
    import alac.net_amd as p

    p.lib()
    return p


class DevBatch:
    def __init__(self, torch, b, misalign=0):
        dev = torch.device("cuda", 0)
        nb = int(b["blob"].size)
        self.raw = torch.zeros((nb + 63) // 16 * 16 + 64 + 16, dtype=torch.uint8, device=dev)
        self.blob = self.raw[misalign:]
        self.blob[:nb] = torch.from_numpy(b["blob"]).to(dev)
        self.nb = nb
        self.off = torch.from_numpy(b["offsets"].astype(np.int64)).to(dev)
        self.sz = torch.from_numpy(b["sizes"].astype(np.int32)).to(dev)
        self.ci = None if b["cfg_idx"] is None else torch.from_numpy(b["cfg_idx"].astype(np.int16)).to(dev)
        self.n = len(b["sizes"])
        self.slot = int(b["slot_ints"])
        self.pcm = torch.zeros((self.n, self.slot), dtype=torch.int32, device=dev)
        self.ob = torch.zeros(self.n, dtype=torch.int32, device=dev)
        self.os = torch.zeros(self.n, dtype=torch.int32, device=dev)
        self.st = torch.full((self.n,), -1, dtype=torch.int32, device=dev)

    def run(self, ctx, stream, ob=True, os_=True):
        ctx.decode_batch_device(self.blob, self.nb, self.off, self.sz, self.ci, self.n, self.pcm, self.slot,
                                self.ob if ob else None, self.os if os_ else None, self.st, stream=stream.cuda_stream)


def _same_pcm(b, got, ref, idx=None):
    """compare what each packet owns of its slot (the rest of a slot is scratch, include/alacgpu.h)"""
    n = len(ref[3])
    idx = np.arange(n) if idx is None else idx
    for p in range(n):
        q = int(idx[p])
        if ref[3][q] == 0:
            ci = 0 if b["cfg_idx"] is None else int(b["cfg_idx"][q])
            cnt = int(ref[2][q]) * int(b["stream_cfgs"][ci][5])
            if not np.array_equal(got[p, :cnt], ref[0][q, :cnt]):
                return False
    return True


def _oracle(oracle, b):
    return oracle.decode_batch(oracle.make_cfgs(b["stream_cfgs"]), b["blob"], b["offsets"], b["sizes"], b["cfg_idx"],
                               b["slot_ints"], n_threads=8)


def test_user_stream_and_null_optional_outputs(torch, pkg, oracle, synth):
    b = synth.make_config_batch(5, n_packets=96)
    ref = _oracle(oracle, b)
    s = torch.cuda.Stream()
    with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
        d = DevBatch(torch, b)
        d.run(ctx, s, ob=False, os_=False)          # NULL d_out_bytes / d_out_samples
        s.synchronize()
        assert np.array_equal(d.st.cpu().numpy(), ref[3])
        assert _same_pcm(b, d.pcm.cpu().numpy(), ref)
        assert (d.ob.cpu().numpy() == 0).all() and (d.os.cpu().numpy() == 0).all()   # untouched
        d.run(ctx, s)
        s.synchronize()
        assert np.array_equal(d.ob.cpu().numpy(), ref[1]) and np.array_equal(d.os.cpu().numpy(), ref[2])
        assert ctx.last_kernel_ms() > 0


def test_bad_arguments_are_refused(torch, pkg, synth):
    b = synth.make_config_batch(2, n_packets=8)
    L = pkg.lib()
    with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
        d = DevBatch(torch, b, misalign=4)          # blob not 16-byte aligned
        with pytest.raises(pkg.AlacGpuError, match="bad argument"):
            d.run(ctx, torch.cuda.current_stream())
        d = DevBatch(torch, b)
        vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        for kill in ("blob", "off", "sz", "pcm", "st"):
            args = dict(blob=vp(d.blob), off=vp(d.off), sz=vp(d.sz), pcm=vp(d.pcm), st=vp(d.st))
            args[kill] = None
            rc = L.alacgpu_decode_batch_device(ctx._ctx, args["blob"], d.nb, args["off"], args["sz"], None, d.n, args["pcm"],
                                               d.slot, None, None, args["st"], None)
            assert rc == -1, kill
        rc = L.alacgpu_decode_batch_device(ctx._ctx, vp(d.blob), d.nb, vp(d.off), vp(d.sz), None, d.n, vp(d.pcm), 0,
                                           None, None, vp(d.st), None)
        assert rc == -1      # slot_ints == 0
        rc = L.alacgpu_decode_batch_device(ctx._ctx, vp(d.blob), d.nb, vp(d.off), vp(d.sz), None, 0, vp(d.pcm), d.slot,
                                           None, None, vp(d.st), None)
        assert rc == 0       # empty batch: nothing to do
        assert L.alacgpu_decode_batch_device(None, vp(d.blob), d.nb, vp(d.off), vp(d.sz), None, d.n, vp(d.pcm), d.slot,
                                             None, None, vp(d.st), None) == -1


def test_calls_in_flight_on_two_streams_share_one_ctx(torch, pkg, oracle, synth):
    # cfg5: (nearly) every group of 8 packets goes through the flag hand-off between the two kernels of a launch pair;
    # two different batches on two streams, issued back to back without waiting, many times over (more launches than
    # the context has launch slots)
    b1 = synth.make_config_batch(5, n_packets=512, seed=1)
    b2 = synth.make_config_batch(5, n_packets=384, seed=2)
    r1, r2 = _oracle(oracle, b1), _oracle(oracle, b2)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with pkg.AlacGpuContext(b1["stream_cfgs"]) as ctx:
        d1, d2 = DevBatch(torch, b1), DevBatch(torch, b2)
        for rep in range(12):
            d1.pcm.zero_(); d2.pcm.zero_(); d1.st.fill_(-1); d2.st.fill_(-1)
            torch.cuda.synchronize()
            d1.run(ctx, s1)
            d2.run(ctx, s2)
            d1.run(ctx, s1)           # the same batch again right behind itself
            s1.synchronize(); s2.synchronize()
            for d, r, bb in ((d1, r1, b1), (d2, r2, b2)):
                assert np.array_equal(d.st.cpu().numpy(), r[3]), rep
                assert _same_pcm(bb, d.pcm.cpu().numpy(), r), rep
                assert np.array_equal(d.ob.cpu().numpy(), r[1])


def test_packet_ranges_of_one_batch_on_several_streams(torch, pkg, oracle, synth):
    # what bench.py's overlapped decode + all-gather does: sub-ranges of one resident batch, one launch pair each
    b = synth.make_config_batch(2, n_packets=200)
    ref = _oracle(oracle, b)
    with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
        d = DevBatch(torch, b)
        streams = [torch.cuda.Stream() for _ in range(3)]
        for k, (lo, hi) in enumerate([(0, 67), (67, 133), (133, 200)]):
            ctx.decode_batch_device(d.blob, d.nb, d.off[lo:hi], d.sz[lo:hi], None, hi - lo, d.pcm[lo:hi], d.slot, d.ob[lo:hi],
                                    d.os[lo:hi], d.st[lo:hi], stream=streams[k].cuda_stream)
        torch.cuda.synchronize()
        assert (d.st.cpu().numpy() == 0).all()
        assert np.array_equal(d.pcm.cpu().numpy(), ref[0])


def test_pinned_buffers_and_chunked_host_path(pkg, oracle, synth):
    # alacgpu_decode_batch cuts batches of 512+ packets into ranges on separate streams; pinned and ordinary memory
    b = synth.make_config_batch(5, n_packets=2600)
    ref = _oracle(oracle, b)
    assert pkg.device_count() >= 1
    with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx, pkg.PinnedBuffer((2600, b["slot_ints"]), np.int32) as pp, \
            pkg.PinnedBuffer(b["blob"].size, np.uint8) as pb:
        pb.array[:] = b["blob"]
        for blob, out in ((b["blob"], None), (pb.array, pp.array)):
            pcm, ob, os_, st = ctx.decode_batch(blob, b["offsets"], b["sizes"], b["cfg_idx"], b["slot_ints"], out=out)
            assert np.array_equal(st, ref[3]) and np.array_equal(ob, ref[1]) and np.array_equal(os_, ref[2])
            assert _same_pcm(b, pcm, ref)
        # packets handed over in shuffled order (offsets no longer ascending): single-upload fallback
        perm = np.random.default_rng(0).permutation(2600)
        pcm, ob, os_, st = ctx.decode_batch(b["blob"], b["offsets"][perm], b["sizes"][perm],
                                            None if b["cfg_idx"] is None else b["cfg_idx"][perm], b["slot_ints"])
        assert np.array_equal(st, ref[3][perm]) and _same_pcm(b, pcm, ref, perm)


def test_page_locked_output_is_written_by_the_kernels_directly(pkg, oracle, synth, monkeypatch):
    # alacgpu_decode_batch with page-locked pcm_out: no download, the kernels store into the caller's memory and channel A is
    # parked in device memory (alac_decode_params::park); ALACGPU_ZERO_COPY=0 takes the copying path.  Same results, both
    # output formats, a slot interior offset (a view into a bigger page-locked buffer), the single-packet entry point.
    b = synth.make_config_batch(5, n_packets=1300)
    ref = _oracle(oracle, b)
    slot = int(b["slot_ints"])
    with pkg.PinnedBuffer((1300 + 3, slot), np.int32) as pp:
        for zc in ("1", "0"):
            monkeypatch.setenv("ALACGPU_ZERO_COPY", zc)
            with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
                pp.array[:] = 0x5A5A5A5A
                view = pp.array[2:1302]                       # starts inside the page-locked allocation
                pcm, ob, os_, st = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], slot, out=view)
                assert np.array_equal(st, ref[3]) and np.array_equal(ob, ref[1]) and np.array_equal(os_, ref[2])
                assert _same_pcm(b, pcm, ref)
                assert (pp.array[:2] == 0x5A5A5A5A).all() and (pp.array[1302:] == 0x5A5A5A5A).all()   # nothing outside the view
                if zc == "1":
                    # nothing but the packet's own samples was written (the parked channel went to device memory)
                    p = int(np.nonzero((ref[3] == 0) & (ref[2] > 0) & (ref[2] < 4096))[0][0])
                    cnt = int(ref[2][p]) * int(b["stream_cfgs"][int(b["cfg_idx"][p])][5])
                    assert (view[p, cnt:] == 0x5A5A5A5A).all()
                ctx.set_output_format(1)
                pcm2, ob2, _, st2 = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], slot, out=view)
                assert np.array_equal(st2, ref[3]) and np.array_equal(ob2, ref[1])
                for p in range(0, 1300, 37):
                    if ref[3][p] != 0:
                        continue
                    cfg = b["stream_cfgs"][int(b["cfg_idx"][p])]
                    bps, cnt = int(cfg[1]) // 8, int(ref[2][p]) * int(cfg[5])
                    v = ref[0][p, :cnt].astype(np.int64)
                    exp = np.stack([(v >> (8 * k)) & 255 for k in range(bps)], axis=1).astype(np.uint8).reshape(-1)
                    assert np.array_equal(pcm2[p].view(np.uint8)[: cnt * bps], exp), p
                ctx.set_output_format(0)
                o0, s0 = int(b["offsets"][0]), int(b["sizes"][0])
                refints, rob, rst = ctx.decode_frame(int(b["cfg_idx"][0]), b["blob"][o0:o0 + s0])
                assert rst == ref[3][0] and rob == ref[1][0]


def test_one_batch_over_several_contexts_from_one_process(pkg, oracle, synth):
    # alacgpu_decode_batch_sharded: how a single-process host uses every GPU of a node (one context per device, one native
    # thread each).  One GPU here, so three contexts on the same device; uneven ranges and a batch smaller than the
    # number of contexts included
    b = synth.make_config_batch(5, n_packets=1501)
    ref = _oracle(oracle, b)
    ctxs = [pkg.AlacGpuContext(b["stream_cfgs"], device=0) for _ in range(3)]
    try:
        pcm, ob, os_, st = pkg.decode_batch_sharded(ctxs, b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], b["slot_ints"])
        assert np.array_equal(st, ref[3]) and np.array_equal(ob, ref[1]) and np.array_equal(os_, ref[2])
        assert _same_pcm(b, pcm, ref)
        b2 = synth.make_config_batch(2, n_packets=2)
        ctx2 = [pkg.AlacGpuContext(b2["stream_cfgs"], device=0) for _ in range(3)]
        r2 = _oracle(oracle, b2)
        g2 = pkg.decode_batch_sharded(ctx2, b2["blob"], b2["offsets"], b2["sizes"], None, b2["slot_ints"])
        assert (g2[3] == 0).all() and np.array_equal(g2[0], r2[0])
        for c in ctx2:
            c.close()
    finally:
        for c in ctxs:
            c.close()


def test_packed_output_through_the_two_range_host_path(pkg, oracle, synth):
    # ALACGPU_OUT_PACKED_LE (FormatSamples fused into the store) on a batch big enough to be cut into two ranges: each range
    # downloads the packed part of its slots with a strided copy; mixed 16 / 24-bit stream cfgs
    b = synth.make_config_batch(5, n_packets=1300)
    ref = _oracle(oracle, b)
    with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
        ctx.set_output_format(1)
        pcm, ob, os_, st = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], b["slot_ints"])
    assert np.array_equal(st, ref[3]) and np.array_equal(ob, ref[1])
    for p in range(1300):
        if ref[3][p] != 0:
            continue
        cfg = b["stream_cfgs"][int(b["cfg_idx"][p])]
        bps, cnt = cfg[1] // 8, int(ref[2][p]) * cfg[5]
        v = ref[0][p, :cnt].astype(np.int64)
        exp = np.stack([(v >> (8 * k)) & 0xFF for k in range(bps)], axis=1).astype(np.uint8).reshape(-1)
        assert np.array_equal(pcm[p].view(np.uint8)[: cnt * bps], exp), p
        assert int(ob[p]) == cnt * bps

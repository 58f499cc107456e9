"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Integer path => the bar is exact equality of every int32 PCM sample, of DecodeFrame's return value,
of the sample count and of the per-packet status.
"""
import numpy as np
import pytest

from bitpack import pack

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import alac.net_amd as p

    p.lib()
    return p


@pytest.fixture(autouse=True, params=["auto", "dense", "ab5", "ab"])
def arrangement(request, monkeypatch):
    """Every test of this module runs four times: with the library's own choice between the builds of the main kernel
    (8 packets per workgroup: 16-step units up to 4096 packets per batch -- what the small test batches get --, 8-step units
    with 128 registers up to 10240, with 96 registers up to 12288; 16 packets per workgroup above), with the 16-packet ("dense")
    arrangement forced, with the 96-register build and with the 128-register 8-step build of the 8-packet one forced
    (ALACGPU_DENSE = 1 / 2 / 4, read when a context is created)."""
    if request.param == "dense":
        monkeypatch.setenv("ALACGPU_DENSE", "1")
    elif request.param == "ab5":
        monkeypatch.setenv("ALACGPU_DENSE", "2")
    elif request.param == "ab":
        monkeypatch.setenv("ALACGPU_DENSE", "4")
    else:
        monkeypatch.delenv("ALACGPU_DENSE", raising=False)
    return request.param


def run_both(pkg, oracle, b, n_threads=8):
    with pkg.AlacGpuContext(b["stream_cfgs"], device=0) as ctx:
        g = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], b["slot_ints"])
    o = oracle.decode_batch(oracle.make_cfgs(b["stream_cfgs"]), b["blob"], b["offsets"], b["sizes"], b["cfg_idx"],
                            b["slot_ints"], n_threads=n_threads)
    return g, o


def assert_same(g, o, cfgs=None, cfg_idx=None):
    gp, gob, gos, gst = g
    op, oob, oos, ost = o
    assert np.array_equal(gst, ost), f"status differs at {np.nonzero(gst != ost)[0][:10]}: {gst[gst != ost][:10]} vs {ost[gst != ost][:10]}"
    assert np.array_equal(gob, oob), "DecodeFrame return value differs"
    assert np.array_equal(gos, oos), "sample count differs"
    ok = np.nonzero(ost == 0)[0]
    for p in ok:
        nc = 2 if cfgs is None else int(cfgs[0 if cfg_idx is None else int(cfg_idx[p])][5])
        cnt = int(oos[p]) * nc
        if not np.array_equal(gp[p, :cnt], op[p, :cnt]):
            bad = np.nonzero(gp[p, :cnt] != op[p, :cnt])[0]
            raise AssertionError(f"packet {p}: {len(bad)} of {cnt} ints differ, first at {bad[:8]}: "
                                 f"gpu {gp[p, bad[:8]]} oracle {op[p, bad[:8]]}")


@pytest.mark.parametrize("cfg,n", [(2, 64), (3, 24), (4, 64), (5, 256)])
def test_baseline_configs_small(pkg, oracle, synth, cfg, n):
    b = synth.make_config_batch(cfg, n_packets=n, want_pcm=True)
    g, o = run_both(pkg, oracle, b)
    assert_same(g, o, b["stream_cfgs"], b["cfg_idx"])
    # and both equal the encoder's source PCM (independent round trip)
    d = b["descs"]
    for p in range(n):
        if o[3][p] == 0:
            nc = int(b["stream_cfgs"][0 if b["cfg_idx"] is None else int(b["cfg_idx"][p])][5])
            ch = 2 if d["stereo"][p] else 1
            cnt = int(d["n"][p]) * ch
            src = b["pcm"][p, :cnt]
            got = g[0][p, : int(d["n"][p]) * nc]
            if ch == nc:
                assert np.array_equal(got, src), f"packet {p} != source PCM"


def test_hand_kats_on_gpu(pkg):
    # the same hand-derived vectors that pin the oracle (tests/test_oracle_kat.py), straight on the GPU
    with pkg.AlacGpuContext([(4096, 16, 40, 10, 14, 2), (4096, 16, 40, 10, 14, 1), (4096, 24, 40, 10, 14, 2)]) as ctx:
        k1 = pack([(3, 1), (4, 0), (12, 0), (1, 1), (2, 0), (1, 1), (32, 2), (16, 1), (16, 0xFFFF), (16, 0x7FFF), (16, 0x8000)])
        k2 = pack([(3, 0), (4, 0), (12, 0), (1, 1), (2, 0), (1, 0), (32, 3), (8, 0), (8, 0), (4, 0), (4, 0), (3, 4), (5, 0),
                   "110", "0", "10", "0"])
        dv = 2 * 0x1234
        k5 = pack([(3, 1), (4, 0), (12, 0), (1, 1), (2, 1), (1, 0), (32, 1), (8, 0), (8, 0), (4, 0), (4, 0), (3, 4), (5, 0),
                   (4, 0), (4, 0), (3, 4), (5, 0), (8, 0xAB), (8, 0xCD), (9, 0x1FF), (17, dv), "0"])
        k9 = pack([(3, 0), (4, 0), (12, 0), (1, 1), (2, 0), (1, 1), (32, 2), (16, 7), (16, 0xFFFE)])
        pk = [k1, k2, k5, k9]
        blob = np.frombuffer(b"".join(pk), dtype=np.uint8)
        sizes = np.array([len(x) for x in pk], dtype=np.uint32)
        offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)  # deliberately unaligned
        pcm, ob, os_, st = ctx.decode_batch(blob, offsets, sizes, np.array([0, 1, 2, 0], dtype=np.uint16), 64)
        assert st.tolist() == [0, 0, 0, 0]
        assert pcm[0, :4].tolist() == [1, -1, 32767, -32768] and ob[0] == 8
        assert pcm[1, :3].tolist() == [1, 0, -1] and ob[1] == 6
        assert pcm[2, :2].tolist() == [0x1234AB, 0xCD] and ob[2] == 6
        assert pcm[3, :4].tolist() == [7, 0, -2, 0] and ob[3] == 8
        ref, out_bytes, s = ctx.decode_frame(2, k5)
        assert s == 0 and out_bytes == 6 and ref[:6].tolist() == [0xAB, 0x34, 0x12, 0xCD, 0, 0]


def test_unaligned_offsets_and_odd_batch(pkg, oracle, synth):
    b = synth.make_config_batch(2, n_packets=7, want_pcm=True)
    # re-pack the packets back to back at odd byte offsets
    parts, offs, pos = [], [], 3
    blob = bytearray(b"\x55" * 3)
    for p in range(7):
        o, s = int(b["offsets"][p]), int(b["sizes"][p])
        offs.append(len(blob))
        blob += bytes(b["blob"][o:o + s]) + b"\xAA" * (p % 3)
    b2 = dict(b)
    b2["blob"] = np.frombuffer(bytes(blob), dtype=np.uint8)
    b2["offsets"] = np.array(offs, dtype=np.uint64)
    g, o = run_both(pkg, oracle, b2)
    assert_same(g, o, b["stream_cfgs"], None)
    assert np.array_equal(g[0], b["pcm"])


def test_statuses(pkg, oracle, synth):
    d = synth.packet_descs(6, n=256, max_samples_per_frame=4096)
    d["channels_field"][0] = 2        # unsupported element
    d["pred_type"][1] = [0, 3]        # unhandled prediction type (stereo B)
    d["pred_order"][2] = [0, 0]       # fine here (n <= 4096)
    d["ub"][3] = 1                    # 16-bit with shift bytes: reference ignores them in Deinterlace16
    d["escape"][4] = 1
    b = synth.make_batch(d, synth.default_signal(7), want_pcm=True)
    b.update(stream_cfgs=[(4096, 16, 40, 10, 14, 2)], cfg_idx=None)
    # truncate the last packet: bitstream overrun
    b["sizes"][5] = b["sizes"][5] // 2
    g, o = run_both(pkg, oracle, b)
    assert o[3].tolist() == [1, 3, 0, 0, 0, 5]
    assert_same(g, o, b["stream_cfgs"], None)


def _random_recipe_batch(synth, seed, count, stereo, is24):
    rng = np.random.default_rng(seed)
    d = synth.packet_descs(count, max_samples_per_frame=4096, sample_size=24 if is24 else 16, stereo=int(stereo))
    d["n"] = rng.integers(1, 900, count)
    d["n"][:4] = [1, 2, 33, 4096]
    d["pred_order"] = rng.integers(0, 32, (count, 2))
    d["quant"] = rng.integers(0, 16, (count, 2))
    d["ricemod"] = rng.integers(0, 8, (count, 2))
    d["mix_shift"] = rng.integers(0, 9, count)
    d["mix_weight"] = np.minimum(rng.integers(0, 256, count), 1 << d["mix_shift"].astype(np.int64))
    d["ub"] = rng.integers(0, 3 if is24 else 1, count)
    rc = rng.random(count) < 0.5
    d["coef_mode"] = np.where(rc, 1, 0)
    d["coefs"] = rng.integers(-3000, 3000, (count, 2, 32))
    d["escape"] = rng.random(count) < 0.05
    return d


@pytest.mark.parametrize("stereo,is24,loud", [(True, False, False), (True, True, True), (False, False, True), (False, True, False)])
def test_random_recipes_vs_oracle(pkg, oracle, synth, stereo, is24, loud):
    d = _random_recipe_batch(synth, 1234 + 2 * stereo + is24, 96, stereo, is24)
    sig = synth.default_signal(77)
    sig["silence_prob"] = 0.3
    if loud:
        sig["amp_lo_log2"], sig["amp_hi_log2"], sig["noise_sigma"] = 14.5, 15.0, 9000.0
    b = synth.make_batch(d, sig, want_pcm=True)
    b.update(stream_cfgs=[(4096, 24 if is24 else 16, 40, 10, 14, 2 if stereo else 1)], cfg_idx=None)
    g, o = run_both(pkg, oracle, b)
    assert (o[3] == 0).all()
    assert_same(g, o, b["stream_cfgs"], None)
    for p in range(len(d)):
        cnt = int(d["n"][p]) * (2 if stereo else 1)
        assert np.array_equal(g[0][p, :cnt], b["pcm"][p, :cnt])


def test_exotic_stream_configs(pkg, oracle, synth):
    # other rice parameters than the usual 40/10/14, several stream configs in one batch
    cfgs = [(4096, 16, 40, 10, 14, 2), (4096, 16, 255, 255, 16, 2), (4096, 16, 8, 0, 1, 2), (4096, 24, 100, 3, 9, 2)]
    d = synth.packet_descs(64, n=700, max_samples_per_frame=4096)
    ci = (np.arange(64) % 4).astype(np.uint16)
    for j, c in enumerate(cfgs):
        m = ci == j
        d["sample_size"][m] = c[1]
        d["rice_history_mult"][m], d["rice_initial_history"][m], d["rice_kmodifier"][m] = c[2], c[3], c[4]
    d["pred_order"] = np.random.default_rng(5).integers(0, 32, (64, 2))
    b = synth.make_batch(d, synth.default_signal(5), want_pcm=True)
    b.update(stream_cfgs=cfgs, cfg_idx=ci)
    g, o = run_both(pkg, oracle, b)
    assert (o[3] == 0).all()
    assert_same(g, o, cfgs, ci)


def test_rice_kmodifier_beyond_sixteen(pkg, oracle, synth):
    # AlacFile.cs:82 takes any byte: a value's k is bounded by the history (<= 16), so kb > 16 only changes the run-length
    # mask (1 << kb) - 1 -- with C#'s shift count masked to five bits (kb 40 -> 0xFF, kb 33 -> 1)
    cfgs = [(4096, 16, 40, 10, 17, 2), (4096, 16, 40, 10, 24, 2), (4096, 16, 40, 10, 33, 2), (4096, 24, 40, 10, 40, 2), (4096, 16, 40, 10, 255, 1)]
    d = synth.packet_descs(80, n=900, max_samples_per_frame=4096)
    ci = (np.arange(80) % 5).astype(np.uint16)
    for j, c in enumerate(cfgs):
        m = ci == j
        d['sample_size'][m] = c[1]
        d['rice_kmodifier'][m] = c[4]
        d['stereo'][m] = 1 if c[5] == 2 else 0
    sig = synth.default_signal(17)
    sig['silence_prob'] = 0.8                      # zero runs: the symbols the mask touches
    sig['silence_min'], sig['silence_max'] = 8, 400
    b = synth.make_batch(d, sig, want_pcm=True)
    b.update(stream_cfgs=cfgs, cfg_idx=ci)
    g, o = run_both(pkg, oracle, b)
    assert (o[3] == 0).all()
    assert_same(g, o, cfgs, ci)
    for p in range(80):
        cnt = int(d['n'][p]) * cfgs[int(ci[p])][5]
        assert np.array_equal(g[0][p, :cnt], b['pcm'][p, :cnt])


def test_mutated_packets_never_hang_and_match(pkg, oracle, synth):
    # flip bits in valid packets; every packet is followed by zero padding so that both decoders see
    # zeros past a (possibly now too short) packet.  The kernel must terminate and agree with the oracle
    # on status; where both say OK the PCM must agree too.
    rng = np.random.default_rng(4321)
    src = synth.make_config_batch(5, n_packets=160, seed=99)
    blob = bytearray()
    offs, sizes = [], []
    for p in range(160):
        o, s = int(src["offsets"][p]), int(src["sizes"][p])
        pkt = bytearray(bytes(src["blob"][o:o + s]))
        for _ in range(int(rng.integers(1, 6))):
            pos = int(rng.integers(3, len(pkt))) if rng.random() < 0.8 else int(rng.integers(0, min(12, len(pkt))))
            pkt[pos] ^= 1 << int(rng.integers(0, 8))
        if rng.random() < 0.2:
            pkt = pkt[: int(rng.integers(4, len(pkt)))]
        offs.append(len(blob))
        sizes.append(len(pkt))
        blob += pkt + bytes(96 * 1024)  # far more zeros than any decoder can run through
    b = dict(src)
    b["blob"] = np.frombuffer(bytes(blob), dtype=np.uint8)
    b["offsets"] = np.array(offs, dtype=np.uint64)
    b["sizes"] = np.array(sizes, dtype=np.uint32)
    g, o = run_both(pkg, oracle, b)
    assert set(np.unique(o[3])) - {0}, "the mutation should break at least some packets"
    assert_same(g, o, b["stream_cfgs"], b["cfg_idx"])


def test_full_size_cfg2_roundtrip_property(pkg, synth):
    # BASELINE configs[1] at full size (4096 packets x 4096 stereo samples): size-independent property
    # decode(encode(pcm)) == pcm, checked without the oracle (which would take minutes single-threaded).
    b = synth.make_config_batch(2, want_pcm=True)
    with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
        pcm, ob, os_, st = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], None, b["slot_ints"])
    assert (st == 0).all() and (os_ == 4096).all() and (ob == 16384).all()
    assert np.array_equal(pcm, b["pcm"])


@pytest.mark.parametrize("count", [4096, 4097, 10240, 10243, 12291])
def test_auto_kernel_choice_at_the_batch_size_thresholds(pkg, oracle, synth, count, monkeypatch):
    # one launch per batch (a single host range), so that the batch size decides: up to 4096 packets the build with 16-step units,
    # up to 10240 the 128-register build of the 8-packet arrangement, 10241..12288 the 96-register one, above that the 16-packet
    # arrangement (alacgpu_api.hip: launch).
    # Short packets keep the oracle fast; ragged sample counts and a last, partly filled workgroup included.
    monkeypatch.setenv("ALACGPU_HOST_CHUNKS", "1")
    stereo = count % 2 == 1
    d = synth.packet_descs(count, n=96, max_samples_per_frame=4096, stereo=int(stereo))
    rng = np.random.default_rng(count)
    d["n"] = rng.integers(1, 129, count)
    d["pred_order"] = rng.integers(1, 9, (count, 2))
    b = synth.make_batch(d, synth.default_signal(11), want_pcm=True)
    cfgs = [(4096, 16, 40, 10, 14, 2 if stereo else 1)]
    o = oracle.decode_batch(oracle.make_cfgs(cfgs), b["blob"], b["offsets"], b["sizes"], None, b["slot_ints"], n_threads=8)
    with pkg.AlacGpuContext(cfgs) as ctx:
        g = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], None, b["slot_ints"])
    assert (o[3] == 0).all()
    assert_same(g, o, cfgs, None)


def test_maximum_frame_length(pkg, oracle, synth):
    # 16384 samples per channel is the reference's scratch size (AlacFile.cs:28): the longest frame it can decode.
    # 24-bit, shift bytes, order 16 and a hassize header: the ring and the bit cursor wrap many times
    d = synth.packet_descs(6, n=16384, max_samples_per_frame=16384, sample_size=24, stereo=1)
    d["n"][:] = [16384, 16384, 16383, 8193, 16384, 1]
    d["ub"][:] = [0, 1, 2, 1, 0, 1]
    d["pred_order"][:, 0] = [16, 8, 31, 4, 30, 8]
    d["pred_order"][:, 1] = [16, 8, 2, 4, 17, 8]
    sig = synth.default_signal(2024)
    sig["silence_prob"] = 0.5
    b = synth.make_batch(d, sig, want_pcm=True)
    b.update(stream_cfgs=[(16384, 24, 40, 10, 14, 2)], cfg_idx=None)
    g, o = run_both(pkg, oracle, b)
    assert (o[3] == 0).all()
    assert_same(g, o, b["stream_cfgs"], None)
    for p in range(6):
        assert np.array_equal(g[0][p, : 2 * int(d["n"][p])], b["pcm"][p, : 2 * int(d["n"][p])])


def test_alacfile_mirror_decode_frame(pkg, oracle, synth):
    # the reference's own call sequence: new AlacFile(samplesize, numchannels); SetInfo(codecData); DecodeFrame(in, out)
    cd = [0] * 24 + [0, 0, 0x10, 0x00, 0, 24, 40, 10, 14, 2, 0, 255, 0, 0, 0x20, 0xE7, 0, 6, 0x9F, 0xE4, 0, 0, 0xAC, 0x44]
    d = synth.packet_descs(1, n=4096, sample_size=24, ub=1, pred_order=16)
    b = synth.make_batch(d, synth.default_signal(3), want_pcm=True)
    pkt = bytes(b["blob"][: int(b["sizes"][0])])
    f = pkg.AlacFile(24, 2)
    f.SetInfo(cd)
    out = np.zeros(1024 * 80, dtype=np.int32)   # AlacContext.cs:65
    nbytes = f.DecodeFrame(pkt, out)
    assert nbytes == 4096 * 3 * 2
    cfg = (4096, 24, 40, 10, 14, 2)
    st_, opcm, ob, n = oracle.decode_frame(cfg, pkt)
    ref = oracle.expand_reference_layout(cfg, opcm, n)
    assert np.array_equal(out[: len(ref)], ref)
    # and FormatSamples gives back the source PCM bytes (little-endian 24-bit)
    got = pkg.format_samples(3, out, nbytes)
    src = b["pcm"][0].astype(np.int32)
    exp = np.stack([src & 0xFF, (src >> 8) & 0xFF, (src >> 16) & 0xFF], axis=1).astype(np.uint8).reshape(-1)
    assert np.array_equal(got, exp)
    f.Dispose()


def test_cpp_host_mirror(pkg, oracle, synth, tmp_path):
    # alac.net_amd/host/AlacFile.hpp: the C++ mirror of AlacFile (ctor / SetInfo / DecodeFrame) over the C ABI
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(pkg.__file__), "host", "alacfile_selftest")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    d = synth.packet_descs(1, n=4096, sample_size=24, ub=1, pred_order=16)
    b = synth.make_batch(d, synth.default_signal(9))
    pkt = bytes(b["blob"][: int(b["sizes"][0])])
    path = tmp_path / "packet.bin"
    path.write_bytes(pkt)
    out = subprocess.run([exe, str(path), "24", "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    cfg = (4096, 24, 40, 10, 14, 2)
    st_, opcm, ob, n = oracle.decode_frame(cfg, pkt)
    ref = oracle.expand_reference_layout(cfg, opcm, n)
    h = 1469598103934665603
    for x in ref.astype(np.uint32).tolist():
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert out.stdout.strip() == f"bytes={ob} ints={len(ref)} fnv={h}", out.stdout
    # error behaviour: sample size 20 -> the reference's exception text
    out = subprocess.run([exe, str(path), "20", "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "FIXME: unimplemented sample size 20" in out.stdout


@pytest.mark.parametrize("cfg", [2, 3, 4, 5])
def test_packed_output_equals_format_samples(pkg, oracle, synth, cfg):
    # ALACGPU_OUT_PACKED_LE: the kernel stores what AlacContext.Read returns (FormatSamples fused, AlacContext.cs:214-256)
    b = synth.make_config_batch(cfg, n_packets=48)
    cfgs = oracle.make_cfgs(b["stream_cfgs"])
    opcm, oob, oos, ost = oracle.decode_batch(cfgs, b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], b["slot_ints"], n_threads=8)
    with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
        ctx.set_output_format(1)
        pcm, ob, os_, st = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], b["slot_ints"])
    assert np.array_equal(st, ost) and np.array_equal(ob, oob)
    raw = pcm.view(np.uint8)
    for p in range(len(st)):
        if ost[p] != 0:
            continue
        ci = 0 if b["cfg_idx"] is None else int(b["cfg_idx"][p])
        row = cfgs[ci:ci + 1]
        ref = oracle.expand_reference_layout(row, opcm[p], int(oos[p]))
        exp = oracle.format_samples(int(row["sample_size"][0]) // 8, ref, int(oob[p]))
        assert np.array_equal(raw[p, : len(exp)], exp), f"packet {p}"


@pytest.mark.parametrize("out_format", [0, 1])
def test_both_two_pass_kernels_share_a_batch(pkg, oracle, synth, out_format):
    # groups of 8 packets alternate between what the main two-pass kernel takes (orders 1..8, or 16 somewhere in the
    # group: two taps per lane) and what it hands to the four-taps-per-lane code of the second launch (order 24, 31 = delta mode, or 0 somewhere
    # in the group); one-channel, uncompressed and short packets mixed in; the last group is partly filled
    count = 8 * 9 + 5
    rng = np.random.default_rng(4711)
    d = synth.packet_descs(count, n=2048, max_samples_per_frame=4096, sample_size=16, stereo=1)
    d["n"] = rng.integers(1, 2049, count)
    d["n"][::2] = 2048
    d["pred_order"] = rng.integers(1, 9, (count, 2))
    for g in range(1, 10, 2):                       # every other group: one packet with a long (or no) predictor
        d["pred_order"][min(8 * g + int(rng.integers(0, 8)), count - 1), int(rng.integers(0, 2))] = [16, 24, 31, 0, 17][g // 2]
    d["stereo"][rng.random(count) < 0.15] = 0       # channels field 0 in a two-channel file
    d["escape"] = rng.random(count) < 0.1
    sig = synth.default_signal(5)
    sig["silence_prob"] = 0.3
    b = synth.make_batch(d, sig, want_pcm=True)
    scfg = [(4096, 16, 40, 10, 14, 2)]
    cfgs = oracle.make_cfgs(scfg)
    o = oracle.decode_batch(cfgs, b["blob"], b["offsets"], b["sizes"], None, b["slot_ints"], n_threads=8)
    assert (o[3] == 0).all()
    with pkg.AlacGpuContext(scfg) as ctx:           # auto: two-pass + fallback
        ctx.set_output_format(out_format)
        g = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], None, b["slot_ints"])
    if out_format == 0:
        assert_same(g, o, scfg, None)
    else:
        assert np.array_equal(g[3], o[3]) and np.array_equal(g[1], o[1]) and np.array_equal(g[2], o[2])
        raw = g[0].view(np.uint8)
        for p in range(count):
            ref = oracle.expand_reference_layout(cfgs[0:1], o[0][p], int(o[2][p]))
            exp = oracle.format_samples(2, ref, int(o[1][p]))
            assert np.array_equal(raw[p, : len(exp)], exp), f"packet {p}"


@pytest.mark.parametrize("stereo", [True, False])
def test_packed_24bit_low_order_streams(pkg, oracle, synth, stereo):
    # 24-bit, LPC orders 1..8 (the layout the two-pass kernel takes), shift bytes 0/1/2, packed output: the two-pass
    # kernel parks channel A in the upper half of the slot while the packed bytes grow from its start
    count = 40
    rng = np.random.default_rng(77 + stereo)
    d = synth.packet_descs(count, n=4096, max_samples_per_frame=4096, sample_size=24, stereo=int(stereo))
    d["n"] = rng.integers(1, 4097, count)
    d["n"][::3] = 4096
    d["pred_order"] = rng.integers(1, 9, (count, 2))
    d["ub"] = rng.integers(0, 3, count)
    d["escape"] = rng.random(count) < 0.1
    d["mix_shift"] = rng.integers(0, 5, count)
    d["mix_weight"] = np.minimum(rng.integers(0, 8, count), 1 << d["mix_shift"].astype(np.int64))
    sig = synth.default_signal(99)
    sig["silence_prob"] = 0.3
    b = synth.make_batch(d, sig, want_pcm=True)
    scfg = [(4096, 24, 40, 10, 14, 2 if stereo else 1)]
    cfgs = oracle.make_cfgs(scfg)
    opcm, oob, oos, ost = oracle.decode_batch(cfgs, b["blob"], b["offsets"], b["sizes"], None, b["slot_ints"], n_threads=8)
    assert (ost == 0).all()
    with pkg.AlacGpuContext(scfg) as ctx:
        ctx.set_output_format(1)
        pcm, ob, os_, st = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], None, b["slot_ints"])
    assert np.array_equal(st, ost) and np.array_equal(ob, oob) and np.array_equal(os_, oos)
    raw = pcm.view(np.uint8)
    for p in range(count):
        ref = oracle.expand_reference_layout(cfgs[0:1], opcm[p], int(oos[p]))
        exp = oracle.format_samples(3, ref, int(oob[p]))
        assert np.array_equal(raw[p, : len(exp)], exp), f"packet {p}"


@pytest.mark.parametrize("cfg", [2, 3])
def test_host_path_reuses_the_callers_array_and_trims_the_packed_copy(pkg, oracle, synth, cfg):
    # decode_batch(out=...) decodes into the caller's array (no fresh allocation); with packed output only the
    # first 2 or 3 bytes per slot int come back over PCIe and the rest of the caller's slot is left alone
    b = synth.make_config_batch(cfg, n_packets=40)
    n, slot = len(b["sizes"]), int(b["slot_ints"])
    bps = max(int(c[1]) // 8 for c in b["stream_cfgs"])
    with pkg.AlacGpuContext(b["stream_cfgs"]) as ctx:
        want = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], slot)
        mine = np.full((n, slot), 0x5A5A5A5A, np.int32)
        got = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], slot, out=mine)
        assert got[0] is mine
        for p in range(n):
            cnt = int(want[2][p]) * int(b["stream_cfgs"][0 if b["cfg_idx"] is None else int(b["cfg_idx"][p])][5])
            assert np.array_equal(mine[p, :cnt], want[0][p, :cnt])
        with pytest.raises(ValueError):
            ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], slot, out=mine[:, :-1])
        ctx.set_output_format(1)
        packed = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], slot)
        mine[:] = 0x5A5A5A5A
        ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], slot, out=mine)
        raw, praw = mine.view(np.uint8), packed[0].view(np.uint8)
        for p in range(n):
            assert np.array_equal(raw[p, : packed[1][p]], praw[p, : packed[1][p]])
        assert (raw[:, bps * slot:] == 0x5A).all()      # beyond what any packet of these cfgs can fill: untouched


@pytest.mark.parametrize("stereo,is24", [(True, False), (True, True), (False, False)])
def test_p8_layout_random(pkg, oracle, synth, stereo, is24):
    # every stream has 1 <= N <= 8, so the split kernels use the 8-lanes-per-stream reconstruction layout;
    # ragged sample counts, escapes, silence and loud content exercise its generic (masked) steps too
    rng = np.random.default_rng(808 + stereo + 2 * is24)
    count = 80
    d = _random_recipe_batch(synth, 99 + stereo, count, stereo, is24)
    d["pred_order"] = rng.integers(1, 9, (count, 2))
    d["n"][:8] = [1, 7, 8, 9, 15, 16, 17, 4096]
    sig = synth.default_signal(4242)
    sig["silence_prob"] = 0.4
    sig["amp_lo_log2"], sig["amp_hi_log2"], sig["noise_sigma"] = 13.0, 15.0, 3000.0
    b = synth.make_batch(d, sig, want_pcm=True)
    b.update(stream_cfgs=[(4096, 24 if is24 else 16, 40, 10, 14, 2 if stereo else 1)], cfg_idx=None)
    g, o = run_both(pkg, oracle, b)
    assert (o[3] == 0).all()
    assert_same(g, o, b["stream_cfgs"], None)


def test_tiny_batches_and_degenerate_packets(pkg, oracle, synth):
    # partially filled workgroups (1, 2, 3, 5 packets) and packets of 0 / 1 / 2 / 5 bytes (everything past the end
    # reads as zeros for both decoders: each packet is followed by zero padding)
    src = synth.make_config_batch(2, n_packets=5, want_pcm=True)
    for npk in (1, 2, 3, 5):
        b = dict(src)
        b["offsets"], b["sizes"] = src["offsets"][:npk], src["sizes"][:npk]
        g, o = run_both(pkg, oracle, b)
        assert (o[3] == 0).all()
        assert_same(g, o, b["stream_cfgs"], None)
        assert np.array_equal(g[0], src["pcm"][:npk])
    blob = bytearray()
    offs, sizes = [], []
    first = bytes(src["blob"][: int(src["sizes"][0])])
    for cut in (0, 1, 2, 5, 40, len(first)):
        offs.append(len(blob))
        sizes.append(cut)
        blob += first[:cut] + bytes(96 * 1024)
    b = dict(src)
    b["blob"] = np.frombuffer(bytes(blob), dtype=np.uint8)
    b["offsets"] = np.array(offs, dtype=np.uint64)
    b["sizes"] = np.array(sizes, dtype=np.uint32)
    g, o = run_both(pkg, oracle, b)
    assert o[3].tolist()[-1] == 0 and all(s != 0 for s in o[3].tolist()[:-1])
    assert_same(g, o, b["stream_cfgs"], None)


@pytest.mark.parametrize("is24", [False, True])
@pytest.mark.parametrize("content", ["loud", "one_silent_stream_in_eight", "very_quiet"])
def test_speculative_tiers_on_full_length_streams(pkg, oracle, synth, is24, content):
    # Full-length packets of equal length keep the entropy wave on its speculative units from the first chunk to the
    # last, so the tier a unit lands on is decided by the content alone:
    #   loud                         escape codes in most units (rice_spec_step_esc; _esc_wide for 24-bit raw values)
    #   one_silent_stream_in_eight   long zero runs in one stream of a wave while the others carry signal (parked streams)
    #   very_quiet                   new run symbols every few samples (units redone by rice_step, run-aware tier)
    count = 32
    rng = np.random.default_rng(5150 + is24)
    d = synth.packet_descs(count, max_samples_per_frame=4096, sample_size=24 if is24 else 16, stereo=1)
    d["n"] = np.full(count, 4096)
    d["pred_order"] = rng.integers(1, 9, (count, 2))
    d["ub"] = rng.integers(0, 3 if is24 else 1, count)
    d["mix_shift"] = rng.integers(0, 4, count)
    d["mix_weight"] = np.minimum(rng.integers(0, 8, count), 1 << d["mix_shift"].astype(np.int64))
    sig = synth.default_signal(31337)
    sig["silence_prob"] = 0.0
    if content == "loud":
        sig["amp_lo_log2"], sig["amp_hi_log2"], sig["noise_sigma"] = 14.0, 15.0, 12000.0
        b = synth.make_batch(d, sig, want_pcm=True)
    elif content == "very_quiet":
        sig["amp_lo_log2"], sig["amp_hi_log2"], sig["noise_sigma"] = 2.0, 5.0, 2.0
        b = synth.make_batch(d, sig, want_pcm=True)
    else:
        silent = synth.default_signal(31337)
        silent["silence_prob"], silent["silence_min"], silent["silence_max"] = 1.0, 3000, 4000
        parts = []
        for p in range(count):     # packet 8k+3 is the silent one of its workgroup
            parts.append(synth.make_batch(d[p:p + 1], silent if p % 8 == 3 else sig, first_index=p, want_pcm=True))
        blob = np.concatenate([x["blob"][int(x["offsets"][0]):int(x["offsets"][0]) + int(x["sizes"][0])] for x in parts] +
                              [np.zeros(16, dtype=np.uint8)])     # slack after the last packet, as make_batch leaves
        sizes = np.array([int(x["sizes"][0]) for x in parts], dtype=np.uint32)
        offsets = np.concatenate([[0], np.cumsum(sizes[:-1], dtype=np.uint64)]).astype(np.uint64)
        b = dict(blob=blob, offsets=offsets, sizes=sizes, slot_ints=parts[0]["slot_ints"],
                 pcm=np.concatenate([x["pcm"] for x in parts]))
    b.update(stream_cfgs=[(4096, 24 if is24 else 16, 40, 10, 14, 2)], cfg_idx=None)
    g, o = run_both(pkg, oracle, b)
    assert (o[3] == 0).all()
    assert_same(g, o, b["stream_cfgs"], None)
    assert np.array_equal(g[0][:, :8192], b["pcm"][:, :8192])


@pytest.mark.parametrize("stereo", [True, False])
@pytest.mark.parametrize("orders", [(1, 9), (9, 17), (17, 32)])
def test_ragged_batches_long_and_short_packets_share_a_workgroup(pkg, oracle, synth, stereo, orders):
    # Most packets are full frames, a few end early at assorted places (chunk boundaries, one sample either side of
    # them, one sample long): in the two-pass kernels a stream that ends becomes a shadow of the longest stream of its wave
    # and the speculative units go on -- in all three FIR arrangements (one tap per lane, two taps per lane, the four-taps / 16-lane
    # kernel with the delta mode), both passes, and for one-channel packets.
    count = 64
    rng = np.random.default_rng(orders[0] * 7 + stereo)
    d = synth.packet_descs(count, max_samples_per_frame=4096, sample_size=16, stereo=int(stereo))
    n = np.full(count, 4096)
    short = [1, 2, 31, 32, 33, 63, 64, 65, 1000, 2047, 2048, 2049, 4064, 4065, 4095, 500]
    n[rng.choice(count, len(short), replace=False)] = short
    n[8:16] = [4096, 4096, 777, 4096, 4096, 4096, 4096, 4096]   # one short packet among seven long ones
    n[16:24] = 1234                                               # a workgroup of equal, short packets
    d["n"] = n
    d["pred_order"] = rng.integers(orders[0], orders[1], (count, 2))
    sig = synth.default_signal(2718)
    sig["silence_prob"] = 0.1
    b = synth.make_batch(d, sig, want_pcm=True)
    b.update(stream_cfgs=[(4096, 16, 40, 10, 14, 2 if stereo else 1)], cfg_idx=None)
    g, o = run_both(pkg, oracle, b)
    assert (o[3] == 0).all()
    assert_same(g, o, b["stream_cfgs"], None)
    for p in range(count):
        cnt = int(n[p]) * (2 if stereo else 1)
        assert np.array_equal(g[0][p, :cnt], b["pcm"][p, :cnt])



def test_hand_kats_round2_on_gpu(pkg):
    # the hand-derived vectors of tests/test_oracle_kat.py (round 2: unsigned mix weight, uncompressed 24-bit, history
    # saturation, run-symbol mask + clz(0) == 40), straight on the GPU, with the expectations written out again
    from test_oracle_kat import kat_escape24_packet, kat_q19_packet, kat_q20_q1_packet, kat_q3_packet

    cfgs = [(4096, 16, 40, 10, 14, 2), (4096, 24, 40, 10, 14, 2), (4096, 24, 40, 10, 14, 1), (4096, 16, 40, 0, 2, 1)]
    pk = [kat_q3_packet(), kat_escape24_packet(), kat_q19_packet(), kat_q20_q1_packet()]
    blob = np.frombuffer(b"".join(pk), dtype=np.uint8)
    sizes = np.array([len(x) for x in pk], dtype=np.uint32)
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    with pkg.AlacGpuContext(cfgs) as ctx:
        pcm, ob, os_, st = ctx.decode_batch(blob, offsets, sizes, np.arange(4, dtype=np.uint16), 64)
    assert st.tolist() == [0, 0, 0, 0]
    assert pcm[0, :2].tolist() == [103, 93] and ob[0] == 4
    assert pcm[1, :2].tolist() == [0x123456, -2] and ob[1] == 6
    assert pcm[2, :2].tolist() == [35000, -3] and ob[2] == 6
    assert pcm[3, :5].tolist() == [0, 0, 0, 0, -1] and ob[3] == 10 and os_[3] == 5
    # KAT-17: a two-channel element in a one-channel stream comes out as its left channel (AlacFile.cs:353-354)
    esc = pack([(3, 1), (4, 0), (12, 0), (1, 1), (2, 0), (1, 1), (32, 2), (16, 0x0001), (16, 0xFFFF), (16, 0x7FFF), (16, 0x8000)])
    pk = [kat_q3_packet(), esc]
    blob = np.frombuffer(b"".join(pk), dtype=np.uint8)
    sizes = np.array([len(x) for x in pk], dtype=np.uint32)
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    with pkg.AlacGpuContext([(4096, 16, 40, 10, 14, 1)]) as ctx:
        pcm, ob, os_, st = ctx.decode_batch(blob, offsets, sizes, None, 64)
    assert st.tolist() == [0, 0] and ob.tolist() == [2, 4]
    assert pcm[0, :1].tolist() == [103] and pcm[1, :2].tolist() == [1, 32767]


def test_mono_element_with_unknown_prediction_type_does_not_throw(pkg, oracle, synth):
    # AlacFile.cs:484-496: a one-channel element whose predictionType is not 0 skips the predictor silently and the output
    # buffer comes out as it is -- behind any compressed frame that is the residual buffer (:486), i.e. the un-predicted
    # residuals; a two-channel element throws (:650,:660).  The host mirrors reproduce the difference.
    d = synth.packet_descs(2, n=64, max_samples_per_frame=4096, stereo=0)
    d["pred_type"][0] = [2, 0]
    d["stereo"][1] = 1
    d["pred_type"][1] = [0, 1]
    b = synth.make_batch(d, synth.default_signal(1))
    pk = [bytes(b["blob"][int(o):int(o) + int(s)]) for o, s in zip(b["offsets"], b["sizes"])]
    cd = [0] * 24 + [0, 0, 0x10, 0x00, 0, 16, 40, 10, 14, 2, 0, 255, 0, 0, 0x20, 0xE7, 0, 6, 0x9F, 0xE4, 0, 0, 0xAC, 0x44]
    f = pkg.AlacFile(16, 2)
    f.SetInfo(cd)
    out = np.full(1024 * 80, 7, dtype=np.int32)
    assert f.DecodeFrame(pk[0], out) == 64 * 4                           # returns outputsize, no exception
    ref = oracle.decode_batch(oracle.make_cfgs([(4096, 16, 40, 10, 14, 2)]), b["blob"], b["offsets"][:1], b["sizes"][:1], None, 8192)
    assert ref[3][0] == 3 and np.array_equal(out[:128], ref[0][0, :128])   # the residuals (L), 0 (R)
    assert len(set(out[0:128:2].tolist())) > 8                           # ... which are not silence
    with pytest.raises(Exception, match="unhandled predicition type"):
        f.DecodeFrame(pk[1], out)
    f.Dispose()

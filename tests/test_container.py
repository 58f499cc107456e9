"""Demuxer -> batch feeder -> AlacContext surface, against files made by the
M4A writer in alac.net_amd/synth/m4a.py (the reference ships no files)."""
import io

import numpy as np
import pytest


def make_file(synth, n_packets=23, sample_size=16, stereo=True, last=1234, seed=5, **kw):
    from alac.net_amd.synth import m4a

    d = synth.packet_descs(n_packets, sample_size=sample_size, stereo=int(stereo), pred_order=8 if sample_size == 16 else 16)
    d["n"][-1] = last
    b = synth.make_batch(d, synth.default_signal(seed), want_pcm=True)
    packets = [bytes(b["blob"][int(o):int(o) + int(s)]) for o, s in zip(b["offsets"], b["sizes"])]
    data = m4a.write_m4a(packets, [int(x) for x in d["n"]], sample_size=sample_size, channels=2 if stereo else 1, **kw)
    ch = 2 if stereo else 1
    pcm = np.concatenate([b["pcm"][p, : int(d["n"][p]) * ch] for p in range(n_packets)])
    return data, packets, pcm, d


def test_demuxer_tables(synth):
    from alac.net_amd import container

    data, packets, pcm, d = make_file(synth)
    res = container.DemuxResT()
    st = container.QtMovieT(container._Stream(io.BytesIO(data)), res).ReadHeader()
    assert st == container.MDAT_OK
    assert (res.SampleSize, res.NumChannels, res.SampleRate) == (16, 2, 44100)
    assert res.SampleByteSize.tolist() == [len(p) for p in packets]
    assert res.TimeToSample == [(22, 4096), (1, 1234)]
    assert res.Stsc == [(1, 5, 1), (5, 3, 1)] and len(res.Stco) == 5
    assert res.MdatLen == sum(len(p) for p in packets)
    # CodecData feeds SetInfo unchanged (AlacFile.cs:63-93)
    import alac.net_amd as pkg
    cfg = pkg.cfg_from_codec_data(res.CodecData[:48], res.SampleSize, res.NumChannels)
    assert [int(cfg[k][0]) for k in ("max_samples_per_frame", "sample_size", "rice_history_mult", "rice_initial_history",
                                     "rice_kmodifier", "num_channels")] == [4096, 16, 40, 10, 14, 2]
    # first chunk offset points at the first packet
    assert data[res.Stco[0]: res.Stco[0] + len(packets[0])] == packets[0]


def test_demuxer_uniform_stsz_and_skipped_atoms(synth):
    # the stsz form with one size for every packet (QTMovieT.cs:575-590) and the atoms the reference skips:
    # top-level free, moov/udta, moov/free, trak/edts (QTMovieT.cs:95-102, :135-177, :668-722)
    from alac.net_amd import container

    data, packets, pcm, d = make_file(synth, n_packets=12, last=4096, uniform_stsz=True, extra_atoms=True)
    res = container.DemuxResT()
    assert container.QtMovieT(container._Stream(io.BytesIO(data)), res).ReadHeader() == container.MDAT_OK
    size = max(len(p) for p in packets)
    assert res.SampleByteSize.tolist() == [size] * 12 and res.MdatLen == 12 * size
    assert res.TimeToSample == [(12, 4096)] and res.Stsc == [(1, 5, 1), (3, 2, 1)]
    assert data[res.Stco[1]: res.Stco[1] + len(packets[5])] == packets[5]      # chunk 2 starts at packet 5


def test_mdat_before_moov_is_rejected_like_the_reference(synth):
    # QTMovieT.cs:746 compares Seek()'s return value (the new position) with 0 -> such files never load
    from alac.net_amd import container

    data, *_ = make_file(synth, n_packets=3, mdat_first=True)
    res = container.DemuxResT()
    assert container.QtMovieT(container._Stream(io.BytesIO(data)), res).ReadHeader() == container.MDAT_CANNOT_SEEK


def test_unknown_atoms_fail(synth):
    from alac.net_amd import container

    data, *_ = make_file(synth, n_packets=2)
    bad = data.replace(b"smhd", b"vmhd")
    assert container.QtMovieT(container._Stream(io.BytesIO(bad)), container.DemuxResT()).ReadHeader() == container.MDAT_NONE


@pytest.mark.gpu
@pytest.mark.parametrize("sample_size,stereo", [(16, True), (24, True), (16, False)])
def test_alaccontext_read_loop_equals_source_pcm(synth, sample_size, stereo):
    # the reference's playback loop: while ((n = ctx.Read(buf)) > 0) consume(buf, n)
    from alac.net_amd import container

    data, packets, pcm, d = make_file(synth, sample_size=sample_size, stereo=stereo)
    with container.AlacContext(io.BytesIO(data), batch_packets=7) as ctx:
        assert ctx.GetNumSamples() == int(d["n"].sum())
        assert (ctx.GetBitsPerSample(), ctx.GetNumChannels(), ctx.GetSampleRate()) == (sample_size, 2 if stereo else 1, 44100)
        buf = np.zeros(1024 * 80, dtype=np.uint8)
        out = bytearray()
        while True:
            n = ctx.Read(buf)
            if n <= 0:
                break
            out += bytes(buf[:n])
        assert ctx.LastSampleNumber == int(d["n"].sum())
    bps = sample_size // 8
    exp = b"".join(int(v).to_bytes(4, "little", signed=True)[:bps] for v in pcm)
    assert bytes(out) == exp


@pytest.mark.gpu
def test_alaccontext_reads_uniform_stsz_file_with_skipped_atoms(synth):
    from alac.net_amd import container

    data, packets, pcm, d = make_file(synth, n_packets=12, last=4096, uniform_stsz=True, extra_atoms=True)
    with container.AlacContext(io.BytesIO(data), batch_packets=5) as ctx:
        buf = np.zeros(1024 * 80, dtype=np.uint8)
        out = bytearray()
        while True:
            n = ctx.Read(buf)
            if n <= 0:
                break
            out += bytes(buf[:n])
        ctx.SetPosition(4096 * 7 + 5)          # a packet in the second chunk: offsets from stco + the uniform size
        n = ctx.Read(buf)
        assert bytes(buf[:n]) == pcm[(4096 * 7 + 5) * 2: 4096 * 8 * 2].astype("<i2").tobytes()
    assert bytes(out) == pcm.astype("<i2").tobytes()


@pytest.mark.gpu
def test_alaccontext_seek(synth):
    from alac.net_amd import container

    data, packets, pcm, d = make_file(synth, n_packets=13, last=4096)
    with container.AlacContext(io.BytesIO(data), batch_packets=4) as ctx:
        buf = np.zeros(1024 * 80, dtype=np.uint8)
        ctx.Read(buf)
        for pos in (4096 * 6 + 100, 17, 4096 * 12 + 4000):
            ctx.SetPosition(pos)
            n = ctx.Read(buf)
            frame = pos // 4096
            # the rest of that frame, starting at the requested sample (16-bit: offset = samples * channels ints)
            exp = pcm[pos * 2:(frame + 1) * 4096 * 2].astype("<i2").tobytes()
            assert n == len(exp) and bytes(buf[:n]) == exp
            # SetPosition sets LastSampleNumber to the END of the frame (AlacContext.cs:283) and the following Read
            # adds the frame's duration once more (:199): the reference's double count, reproduced
            assert ctx.LastSampleNumber == (frame + 2) * 4096


def _fnv(data):
    h = 1469598103934665603
    for x in data:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.gpu
@pytest.mark.parametrize("sample_size", [16, 24])
def test_cpp_alaccontext_mirror(synth, tmp_path, sample_size):
    # alac.net_amd/host/AlacContext.hpp: C++ twin of the demuxer + AlacContext surface, driven like the reference's callers
    import os
    import subprocess
    import alac.net_amd as pkg

    exe = os.path.join(os.path.dirname(pkg.__file__), "host", "alaccontext_selftest")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    data, packets, pcm, d = make_file(synth, n_packets=13, sample_size=sample_size, last=999)
    path = tmp_path / "t.m4a"
    path.write_bytes(data)
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    bps = sample_size // 8
    exp = b"".join(int(v).to_bytes(4, "little", signed=True)[:bps] for v in pcm)
    total = int(d["n"].sum())
    assert out.stdout.strip() == (f"rate=44100 channels=2 bits={sample_size} samples={total} bytes={len(exp)} "
                                  f"fnv={_fnv(exp)} last={total}"), out.stdout
    pos = 4096 * 5 + 321
    out = subprocess.run([exe, str(path), str(pos)], capture_output=True, text=True, timeout=120)
    if sample_size == 16:
        exp = pcm[pos * 2: 6 * 4096 * 2].astype("<i2").tobytes()
    else:
        # the 24-bit seek quirk (see test_alaccontext_seek_24bit_quirk): starts _offset BYTES into the frame
        fb, off = 4096 * 2 * 3, 321 * 2
        packed = b"".join(int(v).to_bytes(4, "little", signed=True)[:3] for v in pcm[5 * 4096 * 2: 6 * 4096 * 2])
        exp = packed[off: off + fb - 3 * off]
    assert out.stdout.strip() == f"seek bytes={len(exp)} fnv={_fnv(exp)} last={7 * 4096}", out.stdout
    # the reference rejects mdat-before-moov files (QTMovieT.cs:746): same exception text as AlacContext.cs:50
    bad, *_ = make_file(synth, n_packets=2, mdat_first=True)
    p2 = tmp_path / "bad.m4a"
    p2.write_bytes(bad)
    out = subprocess.run([exe, str(p2)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "Error while loading the QuickTime movie headers." in out.stdout


@pytest.mark.gpu
def test_seek_past_the_end_is_a_no_op_and_read_goes_on(synth):
    # AlacContext.cs:262-295: a position at or past the end changes nothing; the next Read returns the next packet.
    # With batch prefetch that packet sits decoded in the queue and the stream is ahead: nothing may be dropped.
    from alac.net_amd import container

    data, packets, pcm, d = make_file(synth, n_packets=9, last=4096)
    with container.AlacContext(io.BytesIO(data), batch_packets=4) as ctx:
        buf = np.zeros(1024 * 80, dtype=np.uint8)
        out = bytearray()
        n = ctx.Read(buf)
        out += bytes(buf[:n])
        last = ctx.LastSampleNumber
        ctx.SetPosition(9 * 4096)          # == total samples: not found
        ctx.SetPosition(10**9)
        assert ctx.LastSampleNumber == last
        while True:
            n = ctx.Read(buf)
            if n <= 0:
                break
            out += bytes(buf[:n])
    assert bytes(out) == pcm.astype("<i2").tobytes()


@pytest.mark.gpu
def test_alaccontext_seek_24bit_quirk(synth):
    # after a seek into a frame, _offset counts samples x channels but indexes a buffer that holds one
    # int per BYTE for 24-bit streams, and the byte count drops by _offset x 3 (AlacContext.cs:200-202, :284-286): the
    # bytes handed out start _offset BYTES into the frame and stop 2 x _offset bytes short of its end.  Reproduced.
    from alac.net_amd import container

    data, packets, pcm, d = make_file(synth, n_packets=6, sample_size=24, last=4096)
    packed = b"".join(int(v).to_bytes(4, "little", signed=True)[:3] for v in pcm)
    fb = 4096 * 2 * 3                       # bytes per frame
    with container.AlacContext(io.BytesIO(data), batch_packets=3) as ctx:
        buf = np.zeros(1024 * 80, dtype=np.uint8)
        for pos in (4096 * 3 + 100, 4096 * 1 + 4000, 7):
            ctx.SetPosition(pos)
            n = ctx.Read(buf)
            frame, k = pos // 4096, pos % 4096
            off = k * 2                     # _offset (ints == bytes here)
            assert n == fb - off * 3
            assert bytes(buf[:n]) == packed[frame * fb + off: frame * fb + off + n]


@pytest.mark.gpu
@pytest.mark.parametrize("sample_size", [16, 24])
def test_alacfilereader_mirror_reads_like_a_wavestream(synth, sample_size):
    # AlacNetNAudioAdapter/ALACFileReader.cs: WaveFormat, Length, Read(buffer, offset, count) in odd-sized pieces that
    # straddle packets (the leftover buffer), Position get/set
    from alac.net_amd.naudio_adapter import ALACFileReader

    data, packets, pcm, d = make_file(synth, n_packets=11, sample_size=sample_size, last=777)
    bps = sample_size // 8
    exp = b"".join(int(v).to_bytes(4, "little", signed=True)[:bps] for v in pcm)
    with ALACFileReader(io.BytesIO(data), batch_packets=4) as r:
        wf = r.WaveFormat
        assert (wf.SampleRate, wf.BitsPerSample, wf.Channels, wf.BlockAlign) == (44100, sample_size, 2, 2 * bps)
        assert r.Length == len(exp) and r.Position == 0
        out = bytearray()
        buf = bytearray(20000)
        sizes = [1, 5000, 16384, 3, 17001, 19990]
        k = 0
        while True:
            want = sizes[k % len(sizes)]
            k += 1
            n = r.Read(buf, 7, want)
            if n == 0:
                break
            out += buf[7:7 + n]
            assert n == want or len(out) == len(exp)
        assert bytes(out) == exp
        assert r.Position == r.Length           # LastSampleNumber * BlockAlign at the end of the stream
    if sample_size == 16:
        with ALACFileReader(io.BytesIO(data), batch_packets=4) as r:
            buf = bytearray(70000)
            r.Read(buf, 0, 1000)                # leaves leftovers in the adapter's buffer
            pos_bytes = (4096 * 4 + 10) * 4
            r.Position = pos_bytes              # Program.cs:49 does this from another thread; drops the leftovers
            n = r.Read(buf, 0, 50000)
            assert bytes(buf[:n]) == exp[pos_bytes:pos_bytes + 50000]


@pytest.mark.gpu
def test_cpp_alacfilereader_mirror(synth, tmp_path):
    import os
    import subprocess
    import alac.net_amd as pkg

    exe = os.path.join(os.path.dirname(pkg.__file__), "host", "alaccontext_selftest")
    data, packets, pcm, d = make_file(synth, n_packets=9, sample_size=16, last=555)
    path = tmp_path / "t.m4a"
    path.write_bytes(data)
    exp = pcm.astype("<i2").tobytes()
    total = int(d["n"].sum())
    out = subprocess.run([exe, str(path), "reader", "10007"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip() == (f"reader rate=44100 bits=16 channels=2 align=4 length={len(exp)} bytes={len(exp)} "
                                  f"fnv={_fnv(exp)} position={total * 4}"), out.stdout
    pos = (4096 * 3 + 99) * 4
    out = subprocess.run([exe, str(path), "reader", "9001", str(pos)], capture_output=True, text=True, timeout=120)
    rest = exp[pos:]
    # Position after a seek + reading to the end: the reference's double count of the seek frame (AlacContext.cs:199,:283)
    assert out.stdout.strip() == (f"reader rate=44100 bits=16 channels=2 align=4 length={len(exp)} bytes={len(rest)} "
                                  f"fnv={_fnv(rest)} position={(total + 4096) * 4}"), out.stdout

"""Minimal M4A (MP4) writer for ALAC packets -- test/bench input only (the reference ships no files).

Writes exactly the atom layout the reference's demuxer accepts (QTMovieT.cs): ftyp("M4A "), moov{mvhd,
trak{tkhd, mdia{mdhd, hdlr, minf{smhd(16 bytes), dinf, stbl{stsd{alac}, stts, stsc, stsz, stco}}}}}, mdat --
moov before mdat, one sample description, <= 16 stts entries.
"""
import struct


def _atom(name, payload):
    return struct.pack(">I4s", 8 + len(payload), name.encode("ascii")) + payload


def alac_specific_config(frame_len, sample_size, pb, mb, kb, channels, sample_rate, max_frame_bytes=0, avg_bitrate=0):
    """The 24-byte ALACSpecificConfig (what AlacFile.SetInfo parses at CodecData[24..47])."""
    return struct.pack(">IBBBBBBHIII", frame_len, 0, sample_size, pb, mb, kb, channels, 255, max_frame_bytes, avg_bitrate,
                       sample_rate)


def write_m4a(packets, durations, frame_len=4096, sample_size=16, channels=2, sample_rate=44100, pb=40, mb=10, kb=14,
              packets_per_chunk=5, mdat_first=False, uniform_stsz=False, extra_atoms=False):
    """Returns the file as bytes.  durations[i] = PCM frames in packet i.
    uniform_stsz: every packet is zero-padded to the longest one's size and stsz carries that one size (QTMovieT.cs:575-590).
    extra_atoms: adds the atoms the reference skips -- a top-level `free`, `udta` and `free` inside moov, `edts` inside trak
    (QTMovieT.cs:95-102, :135-177, :668-722)."""
    n = len(packets)
    if uniform_stsz:
        size = max(len(p) for p in packets)
        packets = [p + bytes(size - len(p)) for p in packets]
    cfg = alac_specific_config(frame_len, sample_size, pb, mb, kb, channels, sample_rate)
    inner = _atom("alac", struct.pack(">I", 0) + cfg)                       # size, 'alac', version/flags, config
    entry = (bytes(6) + struct.pack(">HHIH", 1, 0, 0, 0) + struct.pack(">HH", channels, sample_size)
             + struct.pack(">HH", 0, 0) + struct.pack(">I", sample_rate << 16 & 0xFFFFFFFF) + inner)
    stsd = _atom("stsd", struct.pack(">II", 0, 1) + _atom("alac", entry))
    # stts: run-length of durations (the reference holds at most 16 entries)
    runs = []
    for d in durations:
        if runs and runs[-1][1] == d:
            runs[-1][0] += 1
        else:
            runs.append([1, d])
    assert len(runs) <= 16, "the reference's TimeToSample table has 16 entries"
    stts = _atom("stts", struct.pack(">II", 0, len(runs)) + b"".join(struct.pack(">II", c, d) for c, d in runs))
    if uniform_stsz:
        stsz = _atom("stsz", struct.pack(">III", 0, len(packets[0]), n))
    else:
        stsz = _atom("stsz", struct.pack(">III", 0, 0, n) + b"".join(struct.pack(">I", len(p)) for p in packets))
    n_chunks = (n + packets_per_chunk - 1) // packets_per_chunk
    stsc_entries = [(1, packets_per_chunk, 1)]
    if n % packets_per_chunk and n_chunks > 1:
        stsc_entries.append((n_chunks, n % packets_per_chunk, 1))
    stsc = _atom("stsc", struct.pack(">II", 0, len(stsc_entries)) + b"".join(struct.pack(">III", *e) for e in stsc_entries))

    def build(mdat_offset):
        offs, pos = [], mdat_offset + 8
        for c in range(n_chunks):
            offs.append(pos)
            pos += sum(len(p) for p in packets[c * packets_per_chunk:(c + 1) * packets_per_chunk])
        stco = _atom("stco", struct.pack(">II", 0, len(offs)) + b"".join(struct.pack(">I", o) for o in offs))
        stbl = _atom("stbl", stsd + stts + stsc + stsz + stco)
        smhd = _atom("smhd", bytes(8))
        dinf = _atom("dinf", _atom("dref", struct.pack(">II", 0, 1) + _atom("url ", struct.pack(">I", 1))))
        minf = _atom("minf", smhd + dinf + stbl)
        hdlr = _atom("hdlr", struct.pack(">I4s4sIII", 0, b"mhlr", b"soun", 0, 0, 0) + b"\x00")
        mdhd = _atom("mdhd", struct.pack(">IIIIIHH", 0, 0, 0, sample_rate, sum(durations), 0, 0))
        mdia = _atom("mdia", mdhd + hdlr + minf)
        tkhd = _atom("tkhd", bytes(84))
        edts = _atom("edts", _atom("elst", struct.pack(">IIIII", 0, 1, sum(durations), 0, 0x00010000))) if extra_atoms else b""
        trak = _atom("trak", tkhd + edts + mdia)
        mvhd = _atom("mvhd", bytes(100))
        extra = (_atom("udta", _atom("meta", bytes(20))) + _atom("free", bytes(37))) if extra_atoms else b""
        return _atom("moov", mvhd + trak + extra)

    ftyp = _atom("ftyp", b"M4A " + struct.pack(">I", 0) + b"M4A mp42isom")
    if extra_atoms:
        ftyp += _atom("free", bytes(11))
    mdat = _atom("mdat", b"".join(packets))
    if mdat_first:
        moov = build(len(ftyp))
        return ftyp + mdat + moov
    moov = build(0)
    moov = build(len(ftyp) + len(moov))   # offsets depend on moov's (fixed) size
    return ftyp + moov + mdat

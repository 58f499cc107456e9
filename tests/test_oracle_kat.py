"""Hand-derived known-answer tests that pin the CPU oracle.

The reference (teekay/ALAC.NET) ships no tests or fixtures and cannot be run here (C#, no .NET),
so these vectors were derived by hand, line by line, from ALACDecoder/AlacFile.cs.  Each test
names the reference lines it exercises.  They are the oracle's pin: "parity unpinned" by
reference-held fixtures, pinned by these.
"""
import numpy as np

from bitpack import pack

CFG16_ST = (4096, 16, 40, 10, 14, 2)
CFG16_MONO = (4096, 16, 40, 10, 14, 1)


def test_kat1_escape_stereo16(oracle):
    # AlacFile.cs:588-595 (hassize), :665-677 (raw 16-bit, sign-extend), :359-366 (plain interleave)
    pkt = pack([(3, 1), (4, 0), (12, 0), (1, 1), (2, 0), (1, 1), (32, 2),
                (16, 0x0001), (16, 0xFFFF), (16, 0x7FFF), (16, 0x8000)])
    st, pcm, out_bytes, n = oracle.decode_frame(CFG16_ST, pkt)
    assert st == 0 and n == 2 and out_bytes == 8
    assert pcm.tolist() == [1, -1, 32767, -32768]


def test_kat2_rice_zero_run_sign_modifier_mono16(oracle):
    # AlacFile.cs:214-252: k from history (:221-222), value symbol, history update (:229),
    # zero-run branch with k = clz(90)+(106/64)-24 = 2 (:231-249), signModifier on the next value.
    pkt = pack([(3, 0), (4, 0), (12, 0), (1, 1), (2, 0), (1, 0), (32, 3), (8, 0), (8, 0),
                (4, 0), (4, 0), (3, 4), (5, 0), "110", "0", "10", "0"])
    st, pcm, out_bytes, n = oracle.decode_frame(CFG16_MONO, pkt)
    assert st == 0 and n == 3 and out_bytes == 6
    assert pcm.tolist() == [1, 0, -1]
    # the Rice stage alone, on the same bits: residuals [1, 0, -1], 7 bits consumed
    st, res, end = oracle.rice_decode(pack(["110", "0", "10", "0"]), 3, 16, 10, 14, 40)
    assert st == 0 and res.tolist() == [1, 0, -1] and end == 7


def test_kat3_fir_with_adaptation(oracle):
    # AlacFile.cs:284-293 warm-up, :297-311 prediction, :312-332 sign-LMS adaptation (both signs)
    out, coef = oracle.predictor([10, 2, 3, 1, -1], 16, [16, 0], 4)
    assert out.tolist() == [10, 12, 15, 16, 15]
    assert coef.tolist() == [17, 0]


def test_kat3b_fir_special_orders(oracle):
    # N == 0: output = residuals (:261-267); N == 31: first-order delta (:268-282)
    out, _ = oracle.predictor([5, -3, 7], 16, [], 9)
    assert out.tolist() == [5, -3, 7]
    out, _ = oracle.predictor([5, -3, 7, 32767], 16, [0] * 31, 9)
    # 5, 2, 9, sx16(9 + 32767 = 32776) = -32760
    assert out.tolist() == [5, 2, 9, -32760]


def test_kat4_unmix16(oracle):
    # AlacFile.cs:344-355: right = A - ((B*w) >> s), left = right + B; arithmetic >> on negatives
    assert oracle.deinterlace16([100, 100], [10, -10], 2, 2, 3).tolist() == [103, 93, 98, 108]
    assert oracle.deinterlace16([100, 100], [10, -10], 2, 2, 0).tolist() == [100, 10, 100, -10]


def test_kat5_unmix24_shift_bytes(oracle):
    # AlacFile.cs:381-395 via a full packet: 24-bit stereo, ub=1, N=0 both channels, weight 0.
    # A=[0x1234], B=[0]; shift bytes A=0xAB, B=0xCD  ->  l=0x1234AB, r=0x0000CD
    # Rice with rss=17, history 10 -> k=1: dv(A)=2*0x1234=9320 -> escape: nine 1s + 17 raw bits.
    # history := 0xFFFF is not < 128 and n == 1 anyway; B: dv=0 -> bit '0'.
    dv = 2 * 0x1234
    pkt = pack([(3, 1), (4, 0), (12, 0), (1, 1), (2, 1), (1, 0), (32, 1), (8, 0), (8, 0),
                (4, 0), (4, 0), (3, 4), (5, 0), (4, 0), (4, 0), (3, 4), (5, 0),
                (8, 0xAB), (8, 0xCD), (9, 0x1FF), (17, dv), "0"])
    cfg = (4096, 24, 40, 10, 14, 2)
    st, pcm, out_bytes, n = oracle.decode_frame(cfg, pkt)
    assert st == 0 and n == 1 and out_bytes == 6
    assert pcm.tolist() == [0x1234AB, 0x0000CD]
    ref = oracle.expand_reference_layout(cfg, pcm, 1)
    assert ref.tolist() == [0xAB, 0x34, 0x12, 0xCD, 0x00, 0x00]
    assert oracle.format_samples(3, ref, out_bytes).tolist() == [0xAB, 0x34, 0x12, 0xCD, 0x00, 0x00]


def test_kat6_clz_quirk(oracle):
    # AlacFile.cs:170-191: equals clz32 for x != 0, but 40 for x == 0 (:190)
    assert [oracle.clz(x) for x in (0, 1, 3, 90, 0x00FF0000, -1)] == [40, 31, 30, 25, 8, 0]
    for x in (2, 7, 127, 128, 255, 256, 65535, 65536, 0x7FFFFFFF):
        assert oracle.clz(x) == 32 - x.bit_length()


def test_kat7_format_samples_16(oracle):
    # AlacContext.cs:231-242: low 16 bits little-endian, count in BYTES
    assert oracle.format_samples(2, [1, -1, 0x12345], 6).tolist() == [1, 0, 0xFF, 0xFF, 0x45, 0x23]


def test_kat8_set_info(oracle):
    # AlacFile.cs:63-93: 24 skipped bytes, BE32 frame length, then 7A, sampleSize, pb, mb, kb
    cd = [0] * 24 + [0, 0, 0x10, 0x00, 0, 16, 40, 10, 14, 2, 0, 255, 0, 0, 0x20, 0xE7, 0, 6, 0x9F, 0xE4, 0, 0, 0xAC, 0x44]
    cfg = oracle.set_info(cd, 16, 2)
    assert int(cfg["max_samples_per_frame"][0]) == 4096
    assert (int(cfg["sample_size"][0]), int(cfg["rice_history_mult"][0]), int(cfg["rice_initial_history"][0]),
            int(cfg["rice_kmodifier"][0]), int(cfg["num_channels"][0])) == (16, 40, 10, 14, 2)


def test_kat9_mono_in_two_channel_file(oracle):
    # AlacFile.cs:531-541: mono element, file says 2 channels -> L = sample, R = 0; return uses file channels (:436)
    pkt = pack([(3, 0), (4, 0), (12, 0), (1, 1), (2, 0), (1, 1), (32, 2), (16, 7), (16, 0xFFFE)])
    st, pcm, out_bytes, n = oracle.decode_frame(CFG16_ST, pkt)
    assert st == 0 and n == 2 and out_bytes == 8
    assert pcm.tolist() == [7, 0, -2, 0]


def test_kat10_unsupported_element_and_sizes(oracle):
    # channels field 2: nothing decoded, still returns outputsize (:437,:577,:718)
    pkt = pack([(3, 2), (4, 0), (12, 0), (1, 0), (2, 0), (1, 0)], slack=4)
    st, pcm, out_bytes, n = oracle.decode_frame(CFG16_ST, pkt)
    assert st == oracle.ST_UNSUPPORTED_ELEMENT and out_bytes == 4096 * 4
    # sample size 20: "FIXME: unimplemented sample size" (:574,:715)
    pkt = pack([(3, 1), (4, 0), (12, 0), (1, 1), (2, 0), (1, 1), (32, 1), (20, 1), (20, 2)])
    st, _, _, _ = oracle.decode_frame((4096, 20, 40, 10, 14, 2), pkt)
    assert st == oracle.ST_UNSUPPORTED_SAMPLE_SIZE


def test_kat11_rice_escape_and_multi_bit(oracle):
    # EntropyDecodeValue general path (:205-210) with k=3 (history 3000: (3000>>9)+3 = 8 -> clz 28 -> k = 3):
    # symbol '10' + '101': x=1, e=5 -> 1*7 + 4 = 11 -> residual (11+1)/2 = -6 (odd)
    # then history = 3000 + 11*40 - ((3000*40)>>9) = 3000 + 440 - 234 = 3206 -> k = 3 again:
    # symbol '0' + '00' + next bit: e = read 3 bits; here '001' -> e=1 <= 1 -> value 0, un-read one bit.
    # The un-read bit ('1') starts the third symbol: '1' '0' + '010' -> x=1, e=2 -> 7+1 = 8 -> +4
    st, res, end = oracle.rice_decode(pack(["10", "101", "0", "00", "10", "010"]), 3, 16, 3000, 14, 40)
    assert st == 0 and res.tolist() == [-6, 0, 4]
    assert end == 5 + 3 + 5

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


# ---- round 2: more hand-derived vectors for the quirks a shared misreading could hide ----------
# The same packets run on the GPU in tests/test_gpu_parity.py::test_hand_kats_round2_on_gpu.
def kat_q3_packet():
    # Q3: interlacingLeftweight is read UNSIGNED (:600).  Stereo 16-bit, n = 1, order 0 both channels, shift 8, weight 200.
    # rss = 17, history 10 -> k = 1 for the only symbol of each channel; dv(A) = 2 * 100 = 200 and dv(B) = 2 * 10 = 20 are
    # both > 8 -> escape codes: nine 1s + 17 raw bits.  right = 100 - ((10 * 200) >> 8) = 100 - 7 = 93, left = 93 + 10 = 103.
    # (A signed reading, 200 -> -56, would give right = 100 - ((-560) >> 8) = 100 + 3 = 103.)
    return pack([(3, 1), (4, 0), (12, 0), (1, 1), (2, 0), (1, 0), (32, 1), (8, 8), (8, 200),
                 (4, 0), (4, 0), (3, 4), (5, 0), (4, 0), (4, 0), (3, 4), (5, 0),
                 (9, 0x1FF), (17, 200), (9, 0x1FF), (17, 20)])


def kat_escape24_packet():
    # uncompressed 24-bit stereo (:678-693): Readbits(16) << 8 | Readbits(8), then sign-extended from 24 bits
    return pack([(3, 1), (4, 0), (12, 0), (1, 1), (2, 0), (1, 1), (32, 1), (16, 0x1234), (8, 0x56), (16, 0xFFFF), (8, 0xFE)])


def kat_q19_packet():
    # Q19: history becomes 0xFFFF when the decoded value exceeds 0xFFFF (:229) -- not a clamp of the running sum.
    # Mono 24-bit, n = 2, order 0, rss = 24.  Symbol 1: history 10 -> k = 1; escape, raw 70000 -> residual +35000; history :=
    # 0xFFFF.  Symbol 2: (0xFFFF >> 9) + 3 = 130, clz = 24, initialK = 31 - 14 - 24 = -7 -> k = 7: '0' + '0000110' (e = 6) ->
    # dv = 0 * 127 + 5 = 5 -> residual -(6 / 2) = -3.  (With history = 10 + 70000 * 40 the second k would be 14.)
    return pack([(3, 0), (4, 0), (12, 0), (1, 1), (2, 0), (1, 0), (32, 2), (8, 0), (8, 0), (4, 0), (4, 0), (3, 4), (5, 0),
                 (9, 0x1FF), (24, 70000), "0", "0000110"])


def kat_q20_q1_packet():
    # Q20 + Q1: the zero-run symbol is masked with (1 << kmod) - 1 (:236), the value symbol is not (:224); and
    # CountLeadingZeros(0) == 40 makes the run symbol's k = 40 + 0 - 24 = 16 when history is 0 (:234).
    # Stream cfg: initial history 0, k modifier 2; packet: mono 16-bit, n = 5, order 0, rice modifier 0 (history never moves).
    # value: k = 1, '0' -> 0.  run: k = 16, '10' + 16 zero bits: x = 1, e = 0 -> block = 1 * (65535 & 3) = 3, one bit un-read.
    # that bit ('0') is the next value symbol: 0 + signModifier 1 -> dv = 1 -> -1.  4 + 1 < 5 is false: no further run.
    return pack([(3, 0), (4, 0), (12, 0), (1, 1), (2, 0), (1, 0), (32, 5), (8, 0), (8, 0), (4, 0), (4, 0), (3, 0), (5, 0),
                 "0", "10", "0" * 16])


def test_kat12_unsigned_mix_weight(oracle):
    st, pcm, out_bytes, n = oracle.decode_frame(CFG16_ST, kat_q3_packet())
    assert st == 0 and n == 1 and out_bytes == 4 and pcm.tolist() == [103, 93]


def test_kat13_uncompressed_24bit(oracle):
    st, pcm, out_bytes, n = oracle.decode_frame((4096, 24, 40, 10, 14, 2), kat_escape24_packet())
    assert st == 0 and n == 1 and out_bytes == 6 and pcm.tolist() == [0x123456, -2]


def test_kat14_history_saturation(oracle):
    st, pcm, out_bytes, n = oracle.decode_frame((4096, 24, 40, 10, 14, 1), kat_q19_packet())
    assert st == 0 and n == 2 and out_bytes == 6 and pcm.tolist() == [35000, -3]
    st, res, end = oracle.rice_decode(pack([(9, 0x1FF), (24, 70000), "0", "0000110"]), 2, 24, 10, 14, 40)
    assert st == 0 and res.tolist() == [35000, -3] and end == 9 + 24 + 8


def test_kat15_run_symbol_mask_and_clz_of_zero(oracle):
    st, pcm, out_bytes, n = oracle.decode_frame((4096, 16, 40, 0, 2, 1), kat_q20_q1_packet())
    assert st == 0 and n == 5 and out_bytes == 10 and pcm.tolist() == [0, 0, 0, 0, -1]
    st, res, end = oracle.rice_decode(pack(["0", "10", "0" * 16]), 5, 16, 0, 2, 0)
    assert st == 0 and res.tolist() == [0, 0, 0, 0, -1] and end == 19


def test_kat16_order20_fir_both_signs(oracle):
    # order 20, q = 2, coefficient table all zero except coef[19] = 4: warm-up (:284-293) gives out[i] = i + 1 for residuals
    # of 1 (i = 0..20); then i = 21 (base index 0): sum = (out[1] - out[0]) * coef[19] = 1 * 4 -> (2 + 4) >> 2 = 1;
    # out[21] = 1 + out[0] + err.  err = +3: out[21] = 1 + 1 + 3 = 5; adaptation (err > 0), oldest tap first (p = 19):
    # d = out[0] - out[1] = -1, sign -1 -> coef[19] = 5, e -= ((-1 * -1) >> 2) * 1 = 0 -> e stays 3; p = 18: d = out[0] -
    # out[2] = -2 -> coef[18] = 1, e -= (2 >> 2) * 2 = 0; ... p = 16: d = -4 -> coef[16] = 1, e -= (4 >> 2) * 4 = 4 -> e = -1: stop.
    # i = 22 (base index 1): sum = (out[2]-out[1]) * coef[19] + (out[3]-out[1]) * coef[18] + (out[4]-out[1]) * coef[17] +
    #   (out[5]-out[1]) * coef[16] = 1*5 + 2*1 + 3*1 + 4*1 = 14 -> (2 + 14) >> 2 = 4; err = -2: out[22] = 4 + 2 - 2 = 4;
    # adaptation (err < 0): p = 19: d = out[1] - out[2] = -1, sign = -sgn(d) = +1 -> coef[19] = 4; d * sign = -1; e -= (-1 >> 2) * 1
    #   = -(-1) -> e = -1; p = 18: d = -2 -> coef[18] = 0; d * sign = -2 -> (-2 >> 2) = -1, times 2 -> e = -1 + 2 = 1: stop.
    err = [1] * 21 + [3, -2]
    coefs = [0] * 19 + [4]
    out, coef = oracle.predictor(err, 16, coefs, 2)
    assert out.tolist() == list(range(1, 22)) + [5, 4]
    assert coef.tolist() == [0] * 16 + [1, 1, 0, 4]


def test_kat17_two_channel_element_in_a_one_channel_stream(oracle):
    # AlacFile.cs:353-354 with numchannels == 1: out[i] = left, out[i + 1] = right -- every right sample is overwritten by
    # the next left one: the frame comes out as its left channel, the return value counts one channel (:19, :718).
    # KAT-12's packet (left 103, right 93) in a one-channel 16-bit stream:
    st, pcm, out_bytes, n = oracle.decode_frame(CFG16_MONO, kat_q3_packet())
    assert st == 0 and n == 1 and out_bytes == 2 and pcm.tolist() == [103]
    # and two samples, uncompressed: left = [1, 32767], right = [-1, -32768] -> [1, 32767] (then the stray -32768)
    pkt = pack([(3, 1), (4, 0), (12, 0), (1, 1), (2, 0), (1, 1), (32, 2), (16, 0x0001), (16, 0xFFFF), (16, 0x7FFF), (16, 0x8000)])
    st, pcm, out_bytes, n = oracle.decode_frame(CFG16_MONO, pkt)
    assert st == 0 and n == 2 and out_bytes == 4 and pcm.tolist() == [1, 32767]

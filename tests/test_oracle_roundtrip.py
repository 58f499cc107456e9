"""Oracle vs. the independent encoder: decode(encode(pcm)) == pcm under the reference's semantics,
over the parameter space the reference parses (orders 0..31, quantisation, shift bytes, mono/stereo,
ragged sample counts, escape packets)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck


def _check(oracle, synth, d, sig_seed, cfg):
    b = synth.make_batch(d, synth.default_signal(sig_seed), want_pcm=True, n_threads=2)
    pcm, ob, os_, stt = oracle.decode_batch(oracle.make_cfgs([cfg]), b["blob"], b["offsets"], b["sizes"], None, b["slot_ints"])
    assert (stt == 0).all(), stt
    for p in range(len(d)):
        ch = 2 if d["stereo"][p] else 1
        cnt = int(d["n"][p]) * ch
        assert int(os_[p]) == int(d["n"][p])
        assert int(ob[p]) == int(d["n"][p]) * (int(d["sample_size"][p]) // 8) * cfg[5]
        assert np.array_equal(pcm[p, :cnt], b["pcm"][p, :cnt]), f"packet {p} (order {d['pred_order'][p]})"


@pytest.mark.parametrize("cfg", [2, 3, 4, 5])
def test_baseline_config_roundtrip(oracle, synth, cfg):
    n = {2: 24, 3: 8, 4: 24, 5: 96}[cfg]
    b = synth.make_config_batch(cfg, n_packets=n, want_pcm=True, n_threads=4)
    pcm, ob, os_, stt = oracle.decode_batch(oracle.make_cfgs(b["stream_cfgs"]), b["blob"], b["offsets"], b["sizes"],
                                            b["cfg_idx"], b["slot_ints"], n_threads=4)
    assert (stt == 0).all()
    d = b["descs"]
    for p in range(n):
        cnt = int(d["n"][p]) * (2 if d["stereo"][p] else 1)
        assert np.array_equal(pcm[p, :cnt], b["pcm"][p, :cnt])


def test_all_orders_16bit_stereo(oracle, synth):
    d = synth.packet_descs(32, n=300, max_samples_per_frame=4096)
    d["pred_order"][:, 0] = np.arange(32)
    d["pred_order"][:, 1] = np.arange(32)[::-1]
    _check(oracle, synth, d, 11, (4096, 16, 40, 10, 14, 2))


def test_24bit_shift_bytes_and_mono(oracle, synth):
    d = synth.packet_descs(12, n=513, max_samples_per_frame=8192, sample_size=24, pred_order=16)
    d["ub"] = np.arange(12) % 3
    d["stereo"][6:] = 0
    _check(oracle, synth, d[:6], 3, (8192, 24, 40, 10, 14, 2))
    _check(oracle, synth, d[6:], 4, (8192, 24, 40, 10, 14, 1))


def test_digital_silence_hits_zero_run_path(oracle, synth):
    # all-zero PCM: every symbol goes through the "compressed blocks of 0" branch (AlacFile.cs:231-249)
    d = synth.packet_descs(1, n=4096)
    pkt = synth.encode_packet(d[0], np.zeros(8192, dtype=np.int32))
    assert len(pkt) < 200
    st_, pcm, ob, n = oracle.decode_frame((4096, 16, 40, 10, 14, 2), pkt)
    assert st_ == 0 and n == 4096 and not pcm.any()


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(
    seed=st.integers(0, 2**31 - 1),
    n=st.integers(1, 700),
    stereo=st.booleans(),
    is24=st.booleans(),
    na=st.integers(0, 31), nb=st.integers(0, 31),
    q=st.integers(0, 15), ricemod=st.integers(0, 7),
    shift=st.integers(0, 8), weight=st.integers(0, 255),
    random_coefs=st.booleans(), loud=st.booleans(),
)
def test_random_recipes_roundtrip(oracle, synth, seed, n, stereo, is24, na, nb, q, ricemod, shift, weight, random_coefs, loud):
    rng = np.random.default_rng(seed)
    d = synth.packet_descs(1, n=n, max_samples_per_frame=4096, sample_size=24 if is24 else 16, stereo=int(stereo))
    d["pred_order"][0] = [na, nb]
    d["quant"][0] = [q, (q + 3) % 16]
    d["ricemod"][0] = [ricemod, (ricemod + 1) % 8]
    # mixres / 2^mixbits <= 1, as every real encoder keeps it: otherwise the mid channel needs more than
    # sampleSize+1 bits and is not representable in the format at all
    weight = min(weight, 1 << shift)
    d["mix_shift"], d["mix_weight"] = shift, weight
    d["ub"] = int(rng.integers(0, 3 if is24 else 1))
    if random_coefs:
        d["coef_mode"] = 1
        d["coefs"][0] = rng.integers(-3000, 3000, (2, 32))
    sig = synth.default_signal(seed)
    if loud:  # near full scale + wide noise: large residuals, escapes, wrap-around in the predictor
        sig["amp_lo_log2"], sig["amp_hi_log2"], sig["noise_sigma"] = 14.5, 15.0, 9000.0
    b = synth.make_batch(d, sig, want_pcm=True, n_threads=1)
    cfg = (4096, 24 if is24 else 16, 40, 10, 14, 2 if stereo else 1)
    st_, pcm, ob, ns = oracle.decode_frame(cfg, bytes(b["blob"][: int(b["sizes"][0])]))
    cnt = n * (2 if stereo else 1)
    assert st_ == 0 and ns == n
    assert np.array_equal(pcm[:cnt], b["pcm"][0, :cnt])


def test_garbage_never_crashes_the_oracle(oracle):
    rng = np.random.default_rng(99)
    for i in range(300):
        size = int(rng.integers(1, 400))
        pkt = rng.integers(0, 256, size, dtype=np.uint8)
        pkt[0] &= 0x3F  # mostly mono/stereo elements so the decode paths run
        cfg = (int(rng.choice([64, 4096])), int(rng.choice([16, 24])), 40, 10, 14, int(rng.integers(1, 3)))
        st_, pcm, ob, n = oracle.decode_frame(cfg, pkt.tobytes())
        assert 0 <= st_ <= 7

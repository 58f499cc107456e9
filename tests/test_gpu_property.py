"""Property-based differential test: hypothesis draws packet recipes and signal statistics, the
independent encoder makes valid packets, and the HIP path must equal the oracle (and the encoder's source PCM) bit for
bit; then the same packets cut short at a random byte -- WITHOUT zero padding behind them, the next packet follows
immediately -- must give the oracle's status, return value and, where both decode, samples."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu

_CTX = {}


def _ctx(pkg, cfg):
    if cfg not in _CTX:
        _CTX[cfg] = pkg.AlacGpuContext([cfg])
    return _CTX[cfg]


@pytest.fixture(scope="module")
def pkg():
    import alac.net_amd as p

    p.lib()
    yield p
    for c in _CTX.values():
        c.close()
    _CTX.clear()


recipe = st.fixed_dictionaries(dict(
    seed=st.integers(0, 2**31 - 1), count=st.integers(1, 40), stereo=st.booleans(), is24=st.booleans(),
    nmax=st.sampled_from([1, 2, 31, 33, 64, 257, 1000, 4096]), order_hi=st.sampled_from([0, 8, 16, 31]),
    loud=st.sampled_from([0, 1, 2]), silence=st.sampled_from([0.0, 0.3, 1.0]), cut=st.booleans()))


@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(r=recipe)
def test_hypothesis_recipes_gpu_equals_oracle(pkg, oracle, synth, r):
    rng = np.random.default_rng(r["seed"])
    count, stereo, is24 = r["count"], r["stereo"], r["is24"]
    d = synth.packet_descs(count, max_samples_per_frame=4096, sample_size=24 if is24 else 16, stereo=int(stereo))
    d["n"] = rng.integers(1, r["nmax"] + 1, count)
    d["pred_order"] = rng.integers(0, r["order_hi"] + 1, (count, 2))
    d["quant"] = rng.integers(0, 16, (count, 2))
    d["ricemod"] = rng.integers(0, 8, (count, 2))
    d["mix_shift"] = rng.integers(0, 9, count)
    d["mix_weight"] = np.minimum(rng.integers(0, 256, count), 1 << d["mix_shift"].astype(np.int64))
    d["ub"] = rng.integers(0, 3 if is24 else 1, count)
    d["coef_mode"] = rng.integers(0, 2, count)
    d["coefs"] = rng.integers(-3000, 3000, (count, 2, 32))
    d["escape"] = rng.random(count) < 0.05
    sig = synth.default_signal(int(rng.integers(0, 1 << 31)))
    sig["silence_prob"] = r["silence"]
    sig["silence_min"], sig["silence_max"] = 1, 600
    if r["loud"] == 1:
        sig["amp_lo_log2"], sig["amp_hi_log2"], sig["noise_sigma"] = 13.0, 15.0, 6000.0
    elif r["loud"] == 2:
        sig["amp_lo_log2"], sig["amp_hi_log2"], sig["noise_sigma"] = 2.0, 5.0, 2.0
    try:
        b = synth.make_batch(d, sig, want_pcm=True)
    except RuntimeError:
        return     # the encoder refuses a few random recipes (a value it cannot represent)
    cfg = (4096, 24 if is24 else 16, 40, 10, 14, 2 if stereo else 1)
    offsets, sizes = b["offsets"].copy(), b["sizes"].copy()
    if r["cut"]:   # cut some packets short in place: the bytes behind a cut are the rest of the SAME packet / the next one
        for p in range(count):
            if rng.random() < 0.5 and sizes[p] > 4:
                sizes[p] = int(rng.integers(1, sizes[p]))
    o = oracle.decode_batch(oracle.make_cfgs([cfg]), b["blob"], offsets, sizes, None, b["slot_ints"], n_threads=4)
    g = _ctx(pkg, cfg).decode_batch(b["blob"], offsets, sizes, None, b["slot_ints"])
    assert np.array_equal(g[3], o[3]), (g[3], o[3])
    assert np.array_equal(g[1], o[1]) and np.array_equal(g[2], o[2])
    nc = 2 if stereo else 1
    for p in range(count):
        if o[3][p] == 0:
            cnt = int(o[2][p]) * nc
            assert np.array_equal(g[0][p, :cnt], o[0][p, :cnt]), p
            if sizes[p] == b["sizes"][p]:
                assert np.array_equal(g[0][p, :cnt], b["pcm"][p, :cnt]), p

"""Committed golden vectors (tests/golden/golden_v1.npz, made by tests/golden/make_golden.py):
the oracle must keep reproducing them on CPU, and the HIP path must reproduce them on the GPU."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz")


def load():
    z = np.load(G, allow_pickle=False)
    cfgs = [tuple(int(v) for v in row) for row in z["cfgs"]]
    return z, cfgs


def compare(z, cfgs, pcm, ob, os_, st):
    assert st.tolist() == z["status"].tolist()
    assert ob.tolist() == z["out_bytes"].tolist()
    assert os_.tolist() == z["out_samples"].tolist()
    for p in np.nonzero(z["status"] == 0)[0]:
        cnt = int(z["out_samples"][p]) * cfgs[int(z["cfg_idx"][p])][5]
        assert np.array_equal(pcm[p, :cnt], z["pcm"][p, :cnt]), f"golden packet {p}"


def test_oracle_reproduces_golden(oracle):
    z, cfgs = load()
    assert sorted(set(z["status"].tolist())) == [0, 1, 3, 5]
    pcm, ob, os_, st = oracle.decode_batch(oracle.make_cfgs(cfgs), z["blob"], z["offsets"], z["sizes"], z["cfg_idx"],
                                           int(z["slot_ints"]))
    compare(z, cfgs, pcm, ob, os_, st)


@pytest.mark.gpu
def test_gpu_reproduces_golden():
    import alac.net_amd as pkg

    z, cfgs = load()
    with pkg.AlacGpuContext(cfgs) as ctx:
        pcm, ob, os_, st = ctx.decode_batch(z["blob"], z["offsets"], z["sizes"], z["cfg_idx"], int(z["slot_ints"]))
    compare(z, cfgs, pcm, ob, os_, st)

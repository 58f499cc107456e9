"""Committed golden vectors (tests/golden/golden_v2.npz, made by tests/golden/make_golden.py): packets plus what the
LITERAL restatement of AlacFile.cs (oracle/alacfile_literal.py -- a second, independent reading of the reference) returns
for them.  The C oracle (CPU) and the HIP path (GPU) must both reproduce them.  The reference itself ships no fixtures and
cannot run here, so parity remains "unpinned" by reference-held data; these vectors are self-generated, by a different
restatement than the one the parity tests use as their checker."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v2.npz")

# what the reference throws (classified by make_golden.py) -> the per-packet statuses the build may report for it
EXC_TO_STATUS = {2: {2}, 3: {3}, 5: {4, 5}, 6: {6}}


def load():
    z = np.load(G, allow_pickle=False)
    cfgs = [tuple(int(v) for v in row) for row in z["cfgs"]]
    return z, cfgs


def compare(z, cfgs, pcm, ob, os_, st, who):
    checked = 0
    for p in range(len(z["sizes"])):
        cfg = cfgs[int(z["cfg_idx"][p])]
        exc = int(z["ref_exc"][p])
        if exc:
            assert int(st[p]) in EXC_TO_STATUS[exc], f"{who}: packet {p}: reference throws class {exc}, status {st[p]}"
            continue
        # the reference returned normally: same return value; same samples unless the build (by design) refuses to guess
        assert int(ob[p]) == int(z["ref_ret"][p]), f"{who}: packet {p}: DecodeFrame return value"
        if int(st[p]) == 1:      # unknown element: nothing decoded on either side
            continue
        if int(st[p]) == 3:      # one-channel element, unknown prediction type: the reference hands out stale scratch
            assert cfg[5] == 1 or (z["blob"][int(z["offsets"][p])] >> 5) == 0
            continue
        assert int(st[p]) in (0, 5), f"{who}: packet {p}: status {st[p]}"   # 5: cut short (reads on into zeros, like the reference's buffer)
        cnt = int(z["n_samples"][p]) * cfg[5]
        assert int(os_[p]) == int(z["n_samples"][p])
        assert np.array_equal(pcm[p, :cnt], z["pcm"][p, :cnt]), f"{who}: golden packet {p} differs"
        checked += 1
    assert checked >= 36


def test_oracle_reproduces_literal_golden(oracle):
    z, cfgs = load()
    assert sorted(set(z["ref_exc"].tolist())) == [0, 3, 5, 6]
    pcm, ob, os_, st = oracle.decode_batch(oracle.make_cfgs(cfgs), z["blob"], z["offsets"], z["sizes"], z["cfg_idx"],
                                           int(z["slot_ints"]))
    compare(z, cfgs, pcm, ob, os_, st, "C oracle")


def test_literal_restatement_still_gives_the_committed_vectors():
    # the generator's decoder itself, on the three smallest packets that decode (keeps the file and the script together)
    import alacfile_literal as lit

    z, cfgs = load()
    order = [p for p in np.argsort(z["sizes"]) if z["ref_exc"][p] == 0 and z["n_samples"][p] > 0][:3]
    for p in order:
        cfg = cfgs[int(z["cfg_idx"][p])]
        pk = bytes(z["blob"][int(z["offsets"][p]): int(z["offsets"][p]) + int(z["sizes"][p])])
        out, ret = lit.decode_packet(cfg, pk)
        n = int(z["n_samples"][p])
        assert ret == int(z["ref_ret"][p])
        assert lit.canonical_from_reference_layout(out, n, cfg[1], cfg[5]) == z["pcm"][p, : n * cfg[5]].tolist()


@pytest.mark.gpu
def test_gpu_reproduces_literal_golden():
    import alac.net_amd as pkg

    z, cfgs = load()
    with pkg.AlacGpuContext(cfgs) as ctx:
        pcm, ob, os_, st = ctx.decode_batch(z["blob"], z["offsets"], z["sizes"], z["cfg_idx"], int(z["slot_ints"]))
    compare(z, cfgs, pcm, ob, os_, st, "GPU")

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
        if int(st[p]) == 3:      # one-channel element, unknown prediction type: the reference hands out its output buffer as it is --
            # zeros for golden_v2's FRESH decoder (the build reproduces the state behind a normal frame: golden_v3 has that)
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


# ---- golden_v3: BASELINE-shaped, full-length packets through the literal restatement (tests/golden/make_golden_v3.py) ----
G3 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v3.npz")


def load3():
    z = np.load(G3, allow_pickle=False)
    return z, [tuple(int(v) for v in row) for row in z["cfgs"]]


def compare3(z, cfgs, pcm, ob, os_, st, who):
    labels = [str(x) for x in z["labels"]]
    assert len(labels) >= 12 and int(z["n_samples"].max()) == 8192
    for p, label in enumerate(labels):
        cfg = cfgs[int(z["cfg_idx"][p])]
        # the reference returns normally for every one of them; the one-channel element with prediction type 2 carries the
        # build's warning status 3 and the reference's output (the un-predicted residuals, AlacFile.cs:484-496 with :486)
        want_st = 3 if "prediction type 2" in label else 0
        assert int(st[p]) == want_st, f"{who}: {label}: status {st[p]}"
        assert int(ob[p]) == int(z["ref_ret"][p]), f"{who}: {label}: DecodeFrame return value"
        assert int(os_[p]) == int(z["n_samples"][p])
        cnt = int(z["n_samples"][p]) * cfg[5]
        if not np.array_equal(pcm[p, :cnt], z["pcm"][p, :cnt]):
            bad = np.nonzero(pcm[p, :cnt] != z["pcm"][p, :cnt])[0]
            raise AssertionError(f"{who}: {label}: {len(bad)} of {cnt} ints differ, first at {bad[:6]}")


def test_oracle_reproduces_full_length_golden(oracle):
    z, cfgs = load3()
    pcm, ob, os_, st = oracle.decode_batch(oracle.make_cfgs(cfgs), z["blob"], z["offsets"], z["sizes"], z["cfg_idx"], int(z["slot_ints"]))
    compare3(z, cfgs, pcm, ob, os_, st, "C oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("build", ["auto", "dense", "ab5", "ab"])
def test_gpu_reproduces_full_length_golden(build, monkeypatch):
    """All four builds of the first launch (the packets of orders above 8 go through the second launch in every case)."""
    import alac.net_amd as pkg

    if build == "auto":
        monkeypatch.delenv("ALACGPU_DENSE", raising=False)
    else:
        monkeypatch.setenv("ALACGPU_DENSE", {"dense": "1", "ab5": "2", "ab": "4"}[build])
    z, cfgs = load3()
    with pkg.AlacGpuContext(cfgs) as ctx:
        pcm, ob, os_, st = ctx.decode_batch(z["blob"], z["offsets"], z["sizes"], z["cfg_idx"], int(z["slot_ints"]))
        compare3(z, cfgs, pcm, ob, os_, st, f"GPU ({build})")
        # and the same packets in the order that puts every one of them into a different group of 8 with its neighbours shifted
        perm = np.roll(np.arange(len(z["sizes"])), 5)
        pcm2, ob2, os2, st2 = ctx.decode_batch(z["blob"], z["offsets"][perm], z["sizes"][perm], z["cfg_idx"][perm], int(z["slot_ints"]))
    inv = np.argsort(perm)
    compare3(z, cfgs, pcm2[inv], ob2[inv], os2[inv], st2[inv], f"GPU ({build}, rolled)")

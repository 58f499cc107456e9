"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/alacgpu.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "alacgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(alacgpu_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    import alac.net_amd as pkg

    names = declared_symbols()
    assert "alacgpu_decode_batch" in names and "alacgpu_decode_batch_device" in names and len(names) >= 12
    L = pkg.lib()
    for n in names:
        assert hasattr(L, n), f"libalacgpu.so does not export {n}"
        assert n in pkg.SYMBOLS, f"python binding table misses {n}"
    assert L.alacgpu_version() == 3
    assert "alacgpu_comm_create" in names and "alacgpu_allgather_pcm" in names and "alacgpu_shard_ranges" in names


def test_cfg_struct_layout_matches_header():
    import alac.net_amd as pkg

    assert pkg.CFG_DTYPE.itemsize == 12
    assert pkg.CFG_DTYPE.fields["sample_size"][1] == 4 and pkg.CFG_DTYPE.fields["num_channels"][1] == 8


def test_set_info_parse_matches_reference_layout():
    # AlacFile.cs:63-93: 24 skipped bytes, BE32 frame length, 7A, sampleSize, pb, mb, kb ...
    import alac.net_amd as pkg

    cd = [0] * 24 + [0, 0, 0x10, 0x00, 0, 24, 40, 10, 14, 2, 0, 255, 0, 0, 0x20, 0xE7, 0, 6, 0x9F, 0xE4, 0, 0, 0xAC, 0x44]
    cfg = pkg.cfg_from_codec_data(cd, 24, 2)
    assert int(cfg["max_samples_per_frame"][0]) == 4096
    assert [int(cfg[k][0]) for k in ("sample_size", "rice_history_mult", "rice_initial_history", "rice_kmodifier",
                                     "num_channels", "ctor_sample_size")] == [24, 40, 10, 14, 2, 24]


def test_host_side_reshapes_match_oracle(oracle):
    # expand_reference_layout / FormatSamples are host-side steps of the boundary (AlacFile.cs:390-395,
    # AlacContext.cs:214-256); the library's versions must agree with the oracle's restatement.
    import alac.net_amd as pkg

    rng = np.random.default_rng(5)
    pcm = rng.integers(-(1 << 23), 1 << 23, 64).astype(np.int32)
    cfg24 = (4096, 24, 40, 10, 14, 2)
    a = pkg.expand_reference_layout(cfg24, pcm, 32)
    b = oracle.expand_reference_layout(cfg24, pcm, 32)
    assert np.array_equal(a, b) and len(a) == 192
    assert np.array_equal(pkg.format_samples(3, a, 192), oracle.format_samples(3, b, 192))
    pcm16 = rng.integers(-70000, 70000, 64).astype(np.int32)
    assert np.array_equal(pkg.format_samples(2, pcm16, 128), oracle.format_samples(2, pcm16, 128))


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_gpu_fails_loudly_no_fallback():
    import alac.net_amd as pkg

    with pytest.raises(pkg.AlacGpuError):
        pkg.AlacGpuContext([(4096, 16, 40, 10, 14, 2)])
    f = pkg.AlacFile(16, 2)
    cd = [0] * 24 + [0, 0, 0x10, 0x00, 0, 16, 40, 10, 14, 2] + [0] * 14
    with pytest.raises(pkg.AlacGpuError):
        f.SetInfo(cd)


def test_create_rejects_unsupported_config():
    import alac.net_amd as pkg

    L = pkg.lib()
    cfgs = pkg.make_cfgs([(4096, 16, 40, 10, 0, 2)])  # rice_kmodifier 0 (any other byte is taken, as by SetInfo, AlacFile.cs:82)
    ctx = ctypes.c_void_p()
    rc = L.alacgpu_create(cfgs.ctypes.data_as(ctypes.c_void_p), 1, 0, ctypes.byref(ctx))
    assert rc == -4 and not ctx
    assert b"configuration" in L.alacgpu_strerror(rc)


def test_product_does_not_import_the_oracle():
    # the oracle is test infrastructure: nothing under alac.net_amd/ may reference it
    pkgdir = os.path.join(ROOT, "alac.net_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".c", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "alac_oracle" not in txt and "oracle/" not in txt, f"{fn} references the oracle"


def test_product_kernels_carry_no_diagnostics():
    """The product objects are compiled without ALAC_DIAG: after preprocessing, the kernel TU and the C ABI contain no clock
    reads, no stamp pointer, no experiment switches (diagnostics live in alac_diag.h and exist in `make diag` builds only)."""
    import subprocess

    csrc = os.path.join(ROOT, "alac.net_amd", "csrc")
    for src, emits in (("alac_kernels.hip", ("1", "2", "3", "4", "5")), ("alacgpu_api.hip", (None,)), ("alacgpu_comm.hip", (None,))):
        for emit in emits:
            cmd = ["/opt/rocm/bin/hipcc", "-E", "-P", "--offload-arch=gfx950", "--cuda-device-only" if emit else "--cuda-host-only",
                   "-I", os.path.join(ROOT, "include"), "-I", csrc, os.path.join(csrc, src)]
            if emit:
                cmd.insert(2, f"-DALAC_EMIT={emit}")
            r = subprocess.run(cmd, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            # only what comes from this repository's own sources (the HIP headers mention clock64 themselves)
            own = r.stdout[r.stdout.rfind("namespace alacdev"):] if emit else r.stdout[r.stdout.rfind("struct alacgpu_ctx"):] if "api" in src else r.stdout[r.stdout.rfind("struct rccl_api"):]
            assert len(own) > 2000, (src, emit)
            for word in ("clock64", "s_memtime", "dbg", "ALAC_EXPERIMENT", "SpecStats st;\n    st."):
                assert word not in own, f"{src} (ALAC_EMIT={emit}): `{word}` in the product translation unit"
    mk = open(os.path.join(csrc, "Makefile")).read()
    product = mk[:mk.index("diag:")]
    assert "ALAC_DIAG" not in product.replace("# ", "")or all("ALAC_DIAG" not in l for l in product.splitlines() if not l.lstrip().startswith("#"))

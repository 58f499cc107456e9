// alac_diag.h -- diagnostics of the decode kernels.  NOT part of the product: the product objects are compiled without
// ALAC_DIAG, every macro below then expands to nothing, alac_decode_params has no `dbg` member and the kernels contain no
// clock reads (tests/test_abi_and_host.py::test_product_kernels_carry_no_diagnostics preprocesses the product TU and looks).
// `make diag` (-DALAC_DIAG) builds libalacgpu_diag.so for tools/stamps.py: 8 uint64 per workgroup,
//   [0] clock at workgroup start   [2] clock when the entropy wave is done
//   [1] plain units ok << 32 | plain units failed by an escape code
//   [4] zero-run tier units << 32 | escape tier units
//   [5] plain units failed by a new run symbol << 48 | units redone by the generic step << 32
//   [6] full tier units << 48 | late run failures << 32
//   [7] units of the wide plain step << 32 | narrow plain units failed by their range (history / k / value bound)
//   [3] HW_ID of wave 0 | XCC_ID << 32
#ifndef ALAC_DIAG_H
#define ALAC_DIAG_H

#ifdef ALAC_DIAG
struct SpecStats {
    int plain_ok = 0, fail_esc = 0, fail_run = 0, z_units = 0, esc_units = 0, full_units = 0, late_run = 0, redo = 0;
    int fail_range = 0, wide_units = 0;
};
#define SPEC_COUNT(field) (st.field++)
#define DIAG_ONLY(...) __VA_ARGS__
#else
struct SpecStats {};
#define SPEC_COUNT(field) ((void)0)
#define DIAG_ONLY(...)
#endif

#endif

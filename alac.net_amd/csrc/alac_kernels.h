// alac_kernels.h -- launch parameters shared by the kernels and the C-ABI host code.
#ifndef ALAC_KERNELS_H
#define ALAC_KERNELS_H
#include <stdint.h>

// Same layout as alacgpu_cfg (include/alacgpu.h); kept separate so the kernels do not depend on the public header.
struct alacgpu_cfg_dev {
    uint32_t max_samples_per_frame;
    uint8_t sample_size, rice_history_mult, rice_initial_history, rice_kmodifier;
    uint8_t num_channels, ctor_sample_size, reserved, pad;
};
static_assert(sizeof(alacgpu_cfg_dev) == 12, "cfg layout");

// Same numbering as ALACGPU_ST_* (include/alacgpu.h)
enum {
    ALACGPU_ST_OK_D = 0,
    ALACGPU_ST_UNSUPPORTED_ELEMENT_D = 1,
    ALACGPU_ST_UNSUPPORTED_SAMPLE_SIZE_D = 2,
    ALACGPU_ST_UNSUPPORTED_PREDTYPE_D = 3,
    ALACGPU_ST_BAD_SAMPLE_COUNT_D = 4,
    ALACGPU_ST_OVERRUN_D = 5,
    ALACGPU_ST_REF_THROWS_D = 6,
    ALACGPU_ST_UNSUPPORTED_PARAMS_D = 7
};

struct alac_decode_params {
    const uint8_t* blob;        // 16-byte aligned
    uint64_t blob_limit;        // readable bytes from blob (blob_bytes rounded up to 16)
    const uint64_t* offsets;
    const uint32_t* sizes;
    const uint16_t* cfg_idx;    // may be null
    const alacgpu_cfg_dev* cfgs;
    uint32_t n_cfgs;
    uint32_t n_packets;
    int32_t* pcm_out;
    uint32_t slot_ints;
    int32_t* out_bytes;         // may be null
    int32_t* out_samples;       // may be null
    int32_t* status;
    uint32_t out_format;        // 0: one int32 per sample; 1: packed little-endian PCM bytes (FormatSamples fused)
    // Where channel A of a two-channel packet waits between the two passes: null = in the packet's own output slot (ints
    // [n, 2n)); else packet p's place is park + p * park_stride (>= n ints each).  The host-buffer entry point uses a place
    // of its own when pcm_out is page-locked HOST memory the kernels store into directly: the parked samples are read back,
    // and a read across the link costs what the whole decode costs.
    int32_t* park;
    uint32_t park_stride;
#ifdef ALAC_DIAG
    unsigned long long* dbg;    // diagnostic build (make diag) only: 8 stamps per workgroup, see alac_diag.h
#endif
    // alac_decode_ab_kernel, then alac_decode_ab32_kernel: flag g covers packets 8g .. 8g+7.  The first writes 0 where it
    // decoded the group and 1 where it did not; the second decodes the groups flagged 1 and writes 2 there.  One array per
    // launch pair in flight (the host keeps a pool, alacgpu_api.hip: launch_slot).
    uint32_t* ab_flags;
    // Two-pass kernels: one counter per CU (index: XCC_ID, SE_ID, SH_ID, CU_ID), bumped by every workgroup that starts
    // there; its value, the workgroup's turn on the CU, rotates the roles of the workgroup's waves over the SIMDs.
    // 2048 entries, never reset (only the value modulo 4 matters).  Null: roles in wave order.
    uint32_t* cu_arrivals;
};

#ifdef __HIPCC__
// two passes (channel A, then B), 8 packets / 256-thread workgroup (three working waves), LPC orders 1..8
extern "C" __global__ void alac_decode_ab_kernel(alac_decode_params p);
extern "C" __global__ void alac_decode_ab5_kernel(alac_decode_params p);         // the same with 96 registers (five workgroups per CU)
extern "C" __global__ void alac_decode_ab_small_kernel(alac_decode_params p);   // the same with 16-step units (batches up to 4096 packets)
// the second launch, for the groups of 8 packets the first one flagged: two taps per lane of the FIR wave (orders 9..16) or
// four (any order, delta mode, order 0)
extern "C" __global__ void alac_decode_ab32_kernel(alac_decode_params p);
// the first launch with 16 packets / 256-thread workgroup (one entropy wave for 16 streams, orders 1..16): big batches
extern "C" __global__ void alac_decode_ab_dense_kernel(alac_decode_params p);
#endif

#endif

// alacgpu_api.hip -- C ABI of include/alacgpu.h on top of the gfx950 kernels.
// No CPU fallback anywhere in this file: every decode goes through alac_decode_packets_kernel.
#include <hip/hip_runtime.h>
#include <algorithm>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "alac_kernels.h"
#include "alacgpu.h"

static_assert(sizeof(alacgpu_cfg) == sizeof(alacgpu_cfg_dev), "cfg layouts must match");

struct alacgpu_ctx {
    int device = 0;
    uint32_t n_cfgs = 0;
    alacgpu_cfg* h_cfgs = nullptr;
    alacgpu_cfg_dev* d_cfgs = nullptr;
    hipStream_t stream = nullptr;      // used by the host-buffer entry points
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    uint32_t out_format = 0;           // 0 int32 per sample, 1 packed little-endian PCM
    bool all_mono = false;             // every stream cfg has one channel -> the *_mono kernels
    int variant = 0;                   // 0 auto, 1 fused (v1), 2/3/4 split with 1/2/4 reconstruction waves, 5 two-pass
    uint32_t* d_cu_arrivals = nullptr; // two-pass kernels: per-CU workgroup counters (alac_decode_params::cu_arrivals)
    uint32_t* d_ab_flags = nullptr;    // two-pass kernel -> fallback launch protocol (one flag per 8 packets), grow-only
    size_t ab_flags_n = 0;
    // grow-only device workspace for the host-buffer entry points
    void* d_ws = nullptr;
    size_t ws_bytes = 0;
    void* h_pin = nullptr;             // pinned staging for small single-frame calls
    size_t pin_bytes = 0;
    std::string last_error;
};

namespace {

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);              \
            return ALACGPU_ERR_HIP;                                                             \
        }                                                                                       \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int ensure_ws(alacgpu_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return ALACGPU_OK;
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    ctx->d_ws = nullptr;
    ctx->ws_bytes = 0;
    size_t want = align_up(bytes + bytes / 4, 1 << 20);
    HIP_TRY(ctx, hipMalloc(&ctx->d_ws, want));
    ctx->ws_bytes = want;
    return ALACGPU_OK;
}

int launch(alacgpu_ctx* ctx, const alac_decode_params& p_in, hipStream_t stream) {
    if (p_in.n_packets == 0) return ALACGPU_OK;
    alac_decode_params p = p_in;
    p.ab_flags = nullptr;
    p.cu_arrivals = ctx->d_cu_arrivals;
    int variant = ctx->variant;
    // auto: the two-pass kernels (no Rice pre-scan; cfg2 0.99 -> 0.81 ms).  The main one decodes the groups of 8 packets
    // whose streams have LPC order 1..16 and flags the others for the 32-tap arrangement launched right behind it on the
    // same stream.  The split kernels (variants 2..4: 2 / 4 / 8 packets per workgroup, twice that for one-channel cfgs)
    // remain as A/B references, and variant 4 as the choice for one-channel cfgs in very big batches.
    // One-channel cfgs finish in the two-pass kernel's first pass (8 packets per workgroup, no parking); measured on cfg4,
    // two-pass / 16-packet split workgroups: 4096 packets 0.44 / 0.78 ms, 8192 0.57 / 0.81, 12288 0.98 / 0.87,
    // 16384 1.20 / 0.93, 24576 1.60 / 1.91, 32768 2.07 / 2.65: the split kernel wins only where its 1024 workgroups
    // all fit at once and the two-pass kernel's do not.
    if (variant == 0) variant = (ctx->all_mono && p.n_packets > 10240u && p.n_packets <= 20480u) ? 4 : 5;
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, stream));
    if (variant == 5) {
        const size_t groups = ((size_t)p.n_packets + 7) / 8;
        if (groups > ctx->ab_flags_n) {
            if (ctx->d_ab_flags) (void)hipFree(ctx->d_ab_flags);
            ctx->d_ab_flags = nullptr;
            ctx->ab_flags_n = 0;
            const size_t want = groups + groups / 4 + 64;
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_ab_flags, want * sizeof(uint32_t)));
            ctx->ab_flags_n = want;
        }
        p.ab_flags = ctx->d_ab_flags;
        alac_decode_params args = p;
        void* kargs[] = {&args};
        HIP_TRY(ctx, hipLaunchKernel((const void*)alac_decode_ab_kernel, dim3((uint32_t)groups), dim3(256), kargs, 0, stream));
        // what it flagged -> the 32-tap arrangement of the same kernel, which takes every LPC order and so everything that
        // is left (a two-channel packet passes the header check only in a two-channel stream cfg, where its slot has room
        // for parking; the kernel reports a status if that ever fails).  Variants 2..4 remain as A/B references and for
        // one-channel cfgs in very big batches.
        HIP_TRY(ctx, hipLaunchKernel((const void*)alac_decode_ab32_kernel, dim3((uint32_t)groups), dim3(256), kargs, 0, stream));
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));
        ctx->timed = true;
        return ALACGPU_OK;
    }
    // Pick the kernel and its geometry.
    const void* fn = nullptr;
    uint32_t ppw = 2, threads = 64;    // packets per workgroup, workgroup size
    switch (variant) {
    case 1: fn = (const void*)alac_decode_packets_kernel; ppw = 2; threads = 64; break;
    case 2: fn = (const void*)alac_decode_split1_kernel; ppw = 2; threads = 128; break;
    case 4:
        if (ctx->all_mono) { fn = (const void*)alac_decode_split4_mono_kernel; ppw = 16; }
        else { fn = (const void*)alac_decode_split4_kernel; ppw = 8; }
        threads = 320;
        break;
    default:
        if (ctx->all_mono) { fn = (const void*)alac_decode_split2_mono_kernel; ppw = 8; }
        else { fn = (const void*)alac_decode_split2_kernel; ppw = 4; }
        threads = 192;
        break;
    }
    const uint32_t grid = (p.n_packets + ppw - 1) / ppw;
    // (Workgroup placement was checked with HW_ID stamps: a 1024-workgroup grid lands as exactly 4 per CU on all
    // 256 CUs, so no occupancy padding is needed to balance it.)
    const uint32_t dyn_lds = 0;
    {
        alac_decode_params args = p;
        void* kargs[] = {&args};
        HIP_TRY(ctx, hipLaunchKernel(fn, dim3(grid), dim3(threads), kargs, dyn_lds, stream));
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));
    ctx->timed = true;
    return ALACGPU_OK;
}

}  // namespace

extern "C" {

int alacgpu_version(void) { return ALACGPU_VERSION; }

const char* alacgpu_strerror(int rc) {
    switch (rc) {
    case ALACGPU_OK: return "ok";
    case ALACGPU_ERR_BAD_ARG: return "bad argument";
    case ALACGPU_ERR_NO_DEVICE: return "no usable gfx950 device (there is no CPU fallback)";
    case ALACGPU_ERR_HIP: return "HIP runtime error";
    case ALACGPU_ERR_UNSUPPORTED_CONFIG: return "stream configuration outside the supported domain";
    case ALACGPU_ERR_NO_MEMORY: return "out of memory";
    default: return "unknown error";
    }
}

const char* alacgpu_status_string(int st) {
    switch (st) {
    case ALACGPU_ST_OK: return "ok";
    case ALACGPU_ST_UNSUPPORTED_ELEMENT: return "unsupported element (channels field not 0/1)";
    case ALACGPU_ST_UNSUPPORTED_SAMPLE_SIZE: return "FIXME: unimplemented sample size";
    case ALACGPU_ST_UNSUPPORTED_PREDTYPE: return "FIXME: unhandled predicition type";
    case ALACGPU_ST_BAD_SAMPLE_COUNT: return "bad sample count";
    case ALACGPU_ST_OVERRUN: return "bitstream overrun";
    case ALACGPU_ST_REF_THROWS: return "reference throws ArgumentException (order 0, > 4096 samples)";
    case ALACGPU_ST_UNSUPPORTED_PARAMS: return "unsupported parameter combination";
    default: return "unknown status";
    }
}

const char* alacgpu_last_error(alacgpu_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null ctx"; }

int alacgpu_cfg_from_codec_data(const int32_t* in, uint32_t n_ints, int samplesize, int numchannels, alacgpu_cfg* c) {
    if (!in || !c || n_ints < 48) return ALACGPU_ERR_BAD_ARG;
    std::memset(c, 0, sizeof(*c));
    uint32_t p = 24;  // AlacFile.cs:66-71
    c->max_samples_per_frame = ((uint32_t)in[p] << 24) + ((uint32_t)in[p + 1] << 16) + ((uint32_t)in[p + 2] << 8) +
                               (uint32_t)in[p + 3];                       // :72
    c->sample_size = (uint8_t)in[29];                                     // :76
    c->rice_history_mult = (uint8_t)(in[30] & 0xff);                      // :78
    c->rice_initial_history = (uint8_t)(in[31] & 0xff);                   // :80
    c->rice_kmodifier = (uint8_t)(in[32] & 0xff);                         // :82
    c->num_channels = (uint8_t)numchannels;                               // :18
    c->ctor_sample_size = (uint8_t)samplesize;                            // :19
    return ALACGPU_OK;
}

int alacgpu_create(const alacgpu_cfg* cfgs, uint32_t n_cfgs, int device, alacgpu_ctx** out) {
    if (!cfgs || n_cfgs == 0 || !out) return ALACGPU_ERR_BAD_ARG;
    *out = nullptr;
    for (uint32_t i = 0; i < n_cfgs; i++) {
        if (cfgs[i].rice_kmodifier < 1 || cfgs[i].rice_kmodifier > 16) return ALACGPU_ERR_UNSUPPORTED_CONFIG;
        if (cfgs[i].num_channels < 1 || cfgs[i].num_channels > 2) return ALACGPU_ERR_UNSUPPORTED_CONFIG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return ALACGPU_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ALACGPU_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ALACGPU_ERR_NO_DEVICE;  // kernels are gfx950-only
    alacgpu_ctx* ctx = new (std::nothrow) alacgpu_ctx();
    if (!ctx) return ALACGPU_ERR_NO_MEMORY;
    ctx->device = device;
    ctx->n_cfgs = n_cfgs;
    ctx->all_mono = true;
    for (uint32_t i = 0; i < n_cfgs; i++) ctx->all_mono = ctx->all_mono && cfgs[i].num_channels == 1;
    if (const char* v = std::getenv("ALACGPU_KERNEL_VARIANT")) ctx->variant = std::atoi(v) >= 0 && std::atoi(v) <= 5 ? std::atoi(v) : 0;
    int rc = ALACGPU_OK;
    do {
        if (hipSetDevice(device) != hipSuccess) { rc = ALACGPU_ERR_NO_DEVICE; break; }
        ctx->h_cfgs = (alacgpu_cfg*)std::malloc(sizeof(alacgpu_cfg) * n_cfgs);
        if (!ctx->h_cfgs) { rc = ALACGPU_ERR_NO_MEMORY; break; }
        std::memcpy(ctx->h_cfgs, cfgs, sizeof(alacgpu_cfg) * n_cfgs);
        if (hipMalloc((void**)&ctx->d_cfgs, sizeof(alacgpu_cfg_dev) * n_cfgs) != hipSuccess) { rc = ALACGPU_ERR_HIP; break; }
        if (hipMemcpy(ctx->d_cfgs, cfgs, sizeof(alacgpu_cfg) * n_cfgs, hipMemcpyHostToDevice) != hipSuccess) { rc = ALACGPU_ERR_HIP; break; }
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { rc = ALACGPU_ERR_HIP; break; }
        if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) { rc = ALACGPU_ERR_HIP; break; }
        if (hipMalloc((void**)&ctx->d_cu_arrivals, 2048 * sizeof(uint32_t)) != hipSuccess ||
            hipMemset(ctx->d_cu_arrivals, 0, 2048 * sizeof(uint32_t)) != hipSuccess) { rc = ALACGPU_ERR_HIP; break; }
    } while (0);
    if (rc != ALACGPU_OK) {
        alacgpu_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return ALACGPU_OK;
}

void alacgpu_destroy(alacgpu_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->d_cu_arrivals) (void)hipFree(ctx->d_cu_arrivals);
    if (ctx->d_ab_flags) (void)hipFree(ctx->d_ab_flags);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->d_cfgs) (void)hipFree(ctx->d_cfgs);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    std::free(ctx->h_cfgs);
    delete ctx;
}

int alacgpu_decode_batch_device(alacgpu_ctx* ctx, const void* d_blob, uint64_t blob_bytes, const void* d_offsets,
                                const void* d_sizes, const void* d_cfg_idx, uint32_t n_packets, void* d_pcm_out,
                                uint32_t slot_ints, void* d_out_bytes, void* d_out_samples, void* d_status,
                                void* hip_stream) {
    if (!ctx) return ALACGPU_ERR_BAD_ARG;
    if (n_packets == 0) return ALACGPU_OK;
    if (!d_blob || !d_offsets || !d_sizes || !d_pcm_out || !d_status || slot_ints == 0) return ALACGPU_ERR_BAD_ARG;
    if (((uintptr_t)d_blob & 15u) != 0) return ALACGPU_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    alac_decode_params p;
    p.blob = (const uint8_t*)d_blob;
    p.blob_limit = align_up(blob_bytes, 16);
    p.offsets = (const uint64_t*)d_offsets;
    p.sizes = (const uint32_t*)d_sizes;
    p.cfg_idx = (const uint16_t*)d_cfg_idx;
    p.cfgs = ctx->d_cfgs;
    p.n_cfgs = ctx->n_cfgs;
    p.n_packets = n_packets;
    p.pcm_out = (int32_t*)d_pcm_out;
    p.slot_ints = slot_ints;
    p.out_bytes = (int32_t*)d_out_bytes;
    p.out_samples = (int32_t*)d_out_samples;
    p.status = (int32_t*)d_status;
    p.out_format = ctx->out_format;
    p.dbg = nullptr;
    if (std::getenv("ALACGPU_DEBUG_STAMPS")) {  // diagnostic: per-WG phase stamps, printed to stderr (blocks)
        const uint32_t nwg = (n_packets + 1) / 2;
        unsigned long long* d = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&d, sizeof(unsigned long long) * 8 * nwg));
        HIP_TRY(ctx, hipMemset(d, 0, sizeof(unsigned long long) * 8 * nwg));
        p.dbg = d;
        int rc = launch(ctx, p, (hipStream_t)hip_stream);
        HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)hip_stream));
        unsigned long long* h = (unsigned long long*)std::malloc(sizeof(unsigned long long) * 8 * nwg);
        HIP_TRY(ctx, hipMemcpy(h, d, sizeof(unsigned long long) * 8 * nwg, hipMemcpyDeviceToHost));
        double acc[8] = {0}; uint32_t cnt = 0;
        double mx_end = 0, mn_end = 1e30, mx_pre = 0; unsigned long long t_first = ~0ull, t_last = 0;
        for (uint32_t w = 0; w < nwg; w++) {
            if (!h[8 * w]) continue;
            cnt++;
            for (int j = 1; j < 8; j++) acc[j] += (double)(h[8 * w + j] - h[8 * w]);
            const double e = (double)(h[8 * w + 2] - h[8 * w]);
            mx_end = e > mx_end ? e : mx_end; mn_end = e < mn_end ? e : mn_end;
            const double pe = (double)(h[8 * w + 1] - h[8 * w]);
            mx_pre = pe > mx_pre ? pe : mx_pre;
            if (h[8 * w] < t_first) t_first = h[8 * w];
            if (h[8 * w + 2] > t_last) t_last = h[8 * w + 2];
        }
        if (cnt) std::fprintf(stderr, "[alacgpu stamps] per-WG entropy_end min=%.0f max=%.0f  prescan_end max=%.0f  first WG start -> last WG end = %.0f\n",
                              mn_end, mx_end, mx_pre, (double)(t_last - t_first));
        if (cnt) std::fprintf(stderr, "[alacgpu stamps] wgs=%u  prescan_end=%.0f  entropy_end=%.0f  recon_first_chunk=%.0f  recon_end=%.0f  entropy_barrier_wait=%.0f  recon_barrier_wait=%.0f (cycles from WG start)  main_units_redone=%.1f per WG\n",
                              cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, acc[5] / cnt, acc[6] / cnt, acc[7] / cnt);
        if (const char* dump = std::getenv("ALACGPU_DEBUG_STAMPS_FILE")) {   // raw per-WG stamps for offline analysis
            if (FILE* f = std::fopen(dump, "wb")) { std::fwrite(h, sizeof(unsigned long long), (size_t)8 * nwg, f); std::fclose(f); }
        }
        std::free(h); (void)hipFree(d);
        return rc;
    }
    return launch(ctx, p, (hipStream_t)hip_stream);
}

int alacgpu_decode_batch(alacgpu_ctx* ctx, const uint8_t* blob, uint64_t blob_bytes, const uint64_t* offsets,
                         const uint32_t* sizes, const uint16_t* cfg_idx, uint32_t n_packets, int32_t* pcm_out,
                         uint32_t slot_ints, int32_t* out_bytes, int32_t* out_samples, int32_t* status) {
    if (!ctx) return ALACGPU_ERR_BAD_ARG;
    if (n_packets == 0) return ALACGPU_OK;
    if (!blob || !offsets || !sizes || !pcm_out || !status || slot_ints == 0) return ALACGPU_ERR_BAD_ARG;
    for (uint32_t i = 0; i < n_packets; i++)
        if (offsets[i] > blob_bytes || (uint64_t)sizes[i] > blob_bytes - offsets[i]) return ALACGPU_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // workspace carve-up (all 256-byte aligned)
    const size_t blob_sz = align_up(blob_bytes + 64, 256);
    const size_t off_sz = align_up(sizeof(uint64_t) * n_packets, 256);
    const size_t sz_sz = align_up(sizeof(uint32_t) * n_packets, 256);
    const size_t ci_sz = align_up(sizeof(uint16_t) * n_packets, 256);
    const size_t i32_sz = align_up(sizeof(int32_t) * n_packets, 256);
    const size_t pcm_sz = align_up(sizeof(int32_t) * (size_t)n_packets * slot_ints, 256);
    int rc = ensure_ws(ctx, blob_sz + off_sz + sz_sz + ci_sz + 3 * i32_sz + pcm_sz);
    if (rc) return rc;
    uint8_t* w = (uint8_t*)ctx->d_ws;
    uint8_t* d_blob = w; w += blob_sz;
    uint8_t* d_off = w; w += off_sz;
    uint8_t* d_sz = w; w += sz_sz;
    uint8_t* d_ci = w; w += ci_sz;
    uint8_t* d_ob = w; w += i32_sz;
    uint8_t* d_os = w; w += i32_sz;
    uint8_t* d_st = w; w += i32_sz;
    uint8_t* d_pcm = w;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(d_blob, blob, blob_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(d_blob + blob_bytes, 0, blob_sz - blob_bytes, s));
    HIP_TRY(ctx, hipMemcpyAsync(d_off, offsets, sizeof(uint64_t) * n_packets, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(d_sz, sizes, sizeof(uint32_t) * n_packets, hipMemcpyHostToDevice, s));
    if (cfg_idx) HIP_TRY(ctx, hipMemcpyAsync(d_ci, cfg_idx, sizeof(uint16_t) * n_packets, hipMemcpyHostToDevice, s));
    rc = alacgpu_decode_batch_device(ctx, d_blob, blob_bytes, d_off, d_sz, cfg_idx ? d_ci : nullptr, n_packets, d_pcm,
                                     slot_ints, d_ob, d_os, d_st, s);
    if (rc) return rc;
    if (ctx->out_format == ALACGPU_OUT_PACKED_LE) {
        // a slot holds at most slot_ints samples of (ctor sample size / 8) bytes: copy that much of every slot
        size_t bps = 2;
        for (uint32_t i = 0; i < ctx->n_cfgs; i++) {
            const int ss = ctx->h_cfgs[i].ctor_sample_size ? ctx->h_cfgs[i].ctor_sample_size : ctx->h_cfgs[i].sample_size;
            bps = std::max(bps, (size_t)std::min(std::max(ss / 8, 2), 4));
        }
        const size_t pitch = sizeof(int32_t) * (size_t)slot_ints;
        HIP_TRY(ctx, hipMemcpy2DAsync(pcm_out, pitch, d_pcm, pitch, std::min(pitch, bps * (size_t)slot_ints), n_packets,
                                      hipMemcpyDeviceToHost, s));
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(pcm_out, d_pcm, sizeof(int32_t) * (size_t)n_packets * slot_ints, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(ctx, hipMemcpyAsync(status, d_st, sizeof(int32_t) * n_packets, hipMemcpyDeviceToHost, s));
    if (out_bytes) HIP_TRY(ctx, hipMemcpyAsync(out_bytes, d_ob, sizeof(int32_t) * n_packets, hipMemcpyDeviceToHost, s));
    if (out_samples) HIP_TRY(ctx, hipMemcpyAsync(out_samples, d_os, sizeof(int32_t) * n_packets, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return ALACGPU_OK;
}

size_t alacgpu_expand_reference_layout(const alacgpu_cfg* cfg, const int32_t* pcm, int32_t n_samples, int32_t* ref) {
    if (!cfg || !pcm || !ref || n_samples <= 0) return 0;
    const size_t total = (size_t)n_samples * cfg->num_channels;
    if (cfg->sample_size != 24) {
        std::memcpy(ref, pcm, total * sizeof(int32_t));
        return total;
    }
    for (size_t i = 0; i < total; i++) {  // AlacFile.cs:390-395, :555-557
        ref[3 * i + 0] = pcm[i] & 0xFF;
        ref[3 * i + 1] = (pcm[i] >> 8) & 0xFF;
        ref[3 * i + 2] = (pcm[i] >> 16) & 0xFF;
    }
    return 3 * total;
}

size_t alacgpu_format_samples(int bps, const int32_t* src, int32_t samcnt, uint8_t* dst) {  // AlacContext.cs:214-256
    size_t counter = 0, counter2 = 0;
    if (!src || !dst) return 0;
    switch (bps) {
    case 1:
        while (samcnt > 0) { dst[counter] = (uint8_t)(0x00FF & (src[counter] + 128)); counter++; samcnt--; }
        break;
    case 2:
        while (samcnt > 0) {
            int32_t temp = src[counter2];
            dst[counter++] = (uint8_t)temp;
            dst[counter++] = (uint8_t)((uint32_t)temp >> 8);
            counter2++;
            samcnt -= 2;
        }
        break;
    case 3:
        while (samcnt > 0) { dst[counter] = (uint8_t)src[counter2]; counter++; counter2++; samcnt--; }
        break;
    }
    return counter;
}

int alacgpu_decode_frame(alacgpu_ctx* ctx, uint32_t cfg_index, const uint8_t* inbuffer, uint32_t in_bytes,
                         int32_t* outbuffer, uint32_t out_capacity_ints, int32_t* out_bytes, int32_t* status) {
    if (!ctx || !inbuffer || !outbuffer || !status || cfg_index >= ctx->n_cfgs) return ALACGPU_ERR_BAD_ARG;
    const alacgpu_cfg& cfg = ctx->h_cfgs[cfg_index];
    const uint32_t slot = 16384u * cfg.num_channels;
    int32_t* pcm = (int32_t*)std::malloc(sizeof(int32_t) * slot);
    if (!pcm) return ALACGPU_ERR_NO_MEMORY;
    const uint64_t off = 0;
    const uint16_t ci = (uint16_t)cfg_index;
    int32_t ob = 0, os = 0, st = 0;
    const uint32_t saved_format = ctx->out_format;
    ctx->out_format = ALACGPU_OUT_INT32;
    int rc = alacgpu_decode_batch(ctx, inbuffer, in_bytes, &off, &in_bytes, &ci, 1, pcm, slot, &ob, &os, &st);
    ctx->out_format = saved_format;
    if (rc == ALACGPU_OK) {
        *status = st;
        if (out_bytes) *out_bytes = ob;
        if (st == ALACGPU_ST_OK) {
            const size_t need = (size_t)os * cfg.num_channels * (cfg.sample_size == 24 ? 3 : 1);
            if (need > out_capacity_ints) rc = ALACGPU_ERR_BAD_ARG;
            else alacgpu_expand_reference_layout(&cfg, pcm, os, outbuffer);
        }
    }
    std::free(pcm);
    return rc;
}

int alacgpu_set_output_format(alacgpu_ctx* ctx, int format) {
    if (!ctx || (format != ALACGPU_OUT_INT32 && format != ALACGPU_OUT_PACKED_LE)) return ALACGPU_ERR_BAD_ARG;
    ctx->out_format = (uint32_t)format;
    return ALACGPU_OK;
}

int alacgpu_set_kernel_variant(alacgpu_ctx* ctx, int variant) {
    if (!ctx || variant < 0 || variant > 5) return ALACGPU_ERR_BAD_ARG;
    ctx->variant = variant;
    return ALACGPU_OK;
}

float alacgpu_last_kernel_ms(alacgpu_ctx* ctx) {
    if (!ctx || !ctx->timed) return -1.0f;
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0f;
    return ms;
}

}  // extern "C"

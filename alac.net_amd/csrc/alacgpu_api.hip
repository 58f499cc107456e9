// alacgpu_api.hip -- C ABI of include/alacgpu.h on top of the gfx950 kernels.
// No CPU fallback anywhere in this file: every decode goes through alac_decode_ab_kernel / alac_decode_ab32_kernel.
#include <hip/hip_runtime.h>
#include <algorithm>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "alac_kernels.h"
#include "alacgpu.h"

static_assert(sizeof(alacgpu_cfg) == sizeof(alacgpu_cfg_dev), "cfg layouts must match");

namespace {

constexpr int N_SLOTS = 8;          // launch pairs that may be in flight at once on one ctx (any streams)
constexpr uint32_t AB_SMALL_MAX_PACKETS = 4096;  // up to here: the build with 16-step speculative units (latency-bound launches)
constexpr uint32_t AB5_MIN_PACKETS = 10241;     // above: the 96-register build of the 8-packet arrangement (five workgroups per CU)
constexpr uint32_t DENSE_MIN_PACKETS = 12289;   // measured cross-over of the two arrangements of the main kernel
constexpr int N_HOST_STREAMS = 4;   // chunks of the host-buffer pipeline (H2D k+1 || decode k || D2H k-1)

// What one launch pair (alac_decode_ab_kernel + alac_decode_ab32_kernel) owns while it is in flight: the group flags the
// first kernel hands to the second, and the events that bracket the pair.  A slot is reused only after its last launch
// has finished (hipEventSynchronize), so calls on different streams never share flags.
struct launch_slot {
    uint32_t* d_flags = nullptr;
    size_t flags_n = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool used = false;
};

}  // namespace

struct alacgpu_ctx {
    int device = 0;
    uint32_t n_cfgs = 0;
    alacgpu_cfg* h_cfgs = nullptr;
    alacgpu_cfg_dev* d_cfgs = nullptr;
    hipStream_t streams[N_HOST_STREAMS] = {};   // used by the host-buffer entry points: range k decodes (and downloads) on streams[k]
    hipStream_t up_stream = nullptr;            // ... and every upload runs on this one, range after range
    hipEvent_t ev_up[N_HOST_STREAMS] = {};   // range k's packets (and, for k = 0, the batch's metadata) are in HBM
    launch_slot slots[N_SLOTS];
    unsigned next_slot = 0;
    int last_slot = -1;
    uint32_t out_format = 0;           // 0 int32 per sample, 1 packed little-endian PCM
    int host_chunks = 0;               // 0 auto; 1..N_HOST_STREAMS forced (ALACGPU_HOST_CHUNKS, A/B only)
    int dense = -1;                    // main kernel's 16-packet workgroups: -1 auto (by batch size), 0 never, 1 always (ALACGPU_DENSE; A/B and tests)
                                       // 2 / 3 / 4 (ALACGPU_DENSE=..): never, and the 96-register build / the 16-step-unit build / the plain
                                       // 128-register build of the 8-packet arrangement whatever the batch size (tests)
    uint32_t* d_cu_arrivals = nullptr; // per-CU workgroup counters (alac_decode_params::cu_arrivals): ONE array per device, shared by
                                       // every context of the process on it (cu_counters_acquire), so that launches of different
                                       // contexts take their turns on a CU from the same counter
    bool zero_copy = true;             // host-buffer entry points store straight into page-locked output (ALACGPU_ZERO_COPY=0: A/B)
    // grow-only device workspace for the host-buffer entry points
    void* d_ws = nullptr;
    size_t ws_bytes = 0;
    int32_t* h_frame = nullptr;        // pinned staging of alacgpu_decode_frame (one slot of the widest kind)
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

// The per-CU turn counters of a device (see alac_decode_params::cu_arrivals), reference-counted per process: two contexts on
// one GPU -- two-in-flight from two contexts, alacgpu_decode_batch_sharded rehearsed on one device -- must not each believe
// that they alternate the SIMD roles alone.
struct cu_counters { uint32_t* d = nullptr; int refs = 0; };
std::mutex g_cu_mutex;
std::map<int, cu_counters> g_cu_by_device;
uint32_t* cu_counters_acquire(int device) {   // (the caller has made `device` current)
    std::lock_guard<std::mutex> lock(g_cu_mutex);
    cu_counters& c = g_cu_by_device[device];
    if (!c.d) {
        if (hipMalloc((void**)&c.d, 2048 * sizeof(uint32_t)) != hipSuccess) { c.d = nullptr; return nullptr; }
        if (hipMemset(c.d, 0, 2048 * sizeof(uint32_t)) != hipSuccess) { (void)hipFree(c.d); c.d = nullptr; return nullptr; }
    }
    c.refs++;
    return c.d;
}
void cu_counters_release(int device) {
    std::lock_guard<std::mutex> lock(g_cu_mutex);
    auto it = g_cu_by_device.find(device);
    if (it == g_cu_by_device.end()) return;
    if (--it->second.refs <= 0) {
        if (it->second.d) (void)hipFree(it->second.d);
        g_cu_by_device.erase(it);
    }
}

int ensure_ws(alacgpu_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return ALACGPU_OK;
    for (int i = 0; i < N_HOST_STREAMS; i++)
        if (ctx->streams[i]) HIP_TRY(ctx, hipStreamSynchronize(ctx->streams[i]));
    if (ctx->up_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->up_stream));
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    ctx->d_ws = nullptr;
    ctx->ws_bytes = 0;
    size_t want = align_up(bytes + bytes / 4, 1 << 20);
    HIP_TRY(ctx, hipMalloc(&ctx->d_ws, want));
    ctx->ws_bytes = want;
    return ALACGPU_OK;
}

// The two-pass kernels: the first launch decodes the groups of 8 packets whose streams have LPC order 1..8 (the dense
// arrangement: 1..16) and flags the others for the second launch right behind it on the same stream (two or four taps per
// lane of the FIR wave).  A two-channel element needs room for parking channel A in its slot (2 n <= slot_ints): parse_meta
// turns anything else into a per-packet status, also a two-channel element in a one-channel stream cfg, which is decoded
// (its left channel comes out, AlacFile.cs:353-354) when the slot has that room.
int launch(alacgpu_ctx* ctx, const alac_decode_params& p_in, hipStream_t stream) {
    if (p_in.n_packets == 0) return ALACGPU_OK;
    alac_decode_params p = p_in;
    p.cu_arrivals = ctx->d_cu_arrivals;
    const int si = (int)(ctx->next_slot++ % N_SLOTS);
    launch_slot& sl = ctx->slots[si];
    if (!sl.ev0) {   // slots come to life on first use (a context that makes one call at a time only ever touches... all eight, in turn)
        HIP_TRY(ctx, hipEventCreate(&sl.ev0));
        HIP_TRY(ctx, hipEventCreate(&sl.ev1));
    }
    if (sl.used) HIP_TRY(ctx, hipEventSynchronize(sl.ev1));   // the pair that last used these flags has finished
    const size_t groups = ((size_t)p.n_packets + 7) / 8;
    if (groups > sl.flags_n) {
        if (sl.d_flags) (void)hipFree(sl.d_flags);
        sl.d_flags = nullptr;
        sl.flags_n = 0;
        const size_t want = groups + groups / 4 + 64;
        HIP_TRY(ctx, hipMalloc((void**)&sl.d_flags, want * sizeof(uint32_t)));
        sl.flags_n = want;
    }
    p.ab_flags = sl.d_flags;
    HIP_TRY(ctx, hipEventRecord(sl.ev0, stream));
    alac_decode_params args = p;
    void* kargs[] = {&args};
    // Big batches (more workgroups than the chip holds at once: the launch is bound by instruction issue, not by the
    // length of one packet's serial chain) take the dense arrangement: 16 packets per workgroup share one entropy wave.
    const bool dense = ctx->dense < 0 ? p.n_packets >= DENSE_MIN_PACKETS : ctx->dense == 1;
    if (dense)
        HIP_TRY(ctx, hipLaunchKernel((const void*)alac_decode_ab_dense_kernel, dim3((uint32_t)((groups + 1) / 2)), dim3(256), kargs, 0, stream));
    else
    {
        const void* k = (const void*)alac_decode_ab_kernel;
        if (ctx->dense == 2 || (ctx->dense <= 0 && p.n_packets >= AB5_MIN_PACKETS)) k = (const void*)alac_decode_ab5_kernel;
        else if (ctx->dense == 3 || (ctx->dense <= 0 && p.n_packets <= AB_SMALL_MAX_PACKETS)) k = (const void*)alac_decode_ab_small_kernel;
        HIP_TRY(ctx, hipLaunchKernel(k, dim3((uint32_t)groups), dim3(256), kargs, 0, stream));
    }
    HIP_TRY(ctx, hipLaunchKernel((const void*)alac_decode_ab32_kernel, dim3((uint32_t)groups), dim3(256), kargs, 0, stream));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(sl.ev1, stream));
    sl.used = true;
    ctx->last_slot = si;
    return ALACGPU_OK;
}

int fill_params(alacgpu_ctx* ctx, alac_decode_params& p, const void* d_blob, uint64_t blob_bytes, const void* d_offsets,
                const void* d_sizes, const void* d_cfg_idx, uint32_t n_packets, void* d_pcm_out, uint32_t slot_ints,
                void* d_out_bytes, void* d_out_samples, void* d_status) {
    if (!d_blob || !d_offsets || !d_sizes || !d_pcm_out || !d_status || slot_ints == 0) return ALACGPU_ERR_BAD_ARG;
    if (((uintptr_t)d_blob & 15u) != 0 || ((uintptr_t)d_offsets & 7u) != 0 || ((uintptr_t)d_sizes & 3u) != 0 ||
        ((uintptr_t)d_pcm_out & 3u) != 0 || ((uintptr_t)d_status & 3u) != 0 || ((uintptr_t)d_cfg_idx & 1u) != 0 ||
        ((uintptr_t)d_out_bytes & 3u) != 0 || ((uintptr_t)d_out_samples & 3u) != 0)
        return ALACGPU_ERR_BAD_ARG;
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
#ifdef ALAC_DIAG
    p.dbg = nullptr;
#endif
    p.ab_flags = nullptr;
    p.cu_arrivals = nullptr;
    p.park = nullptr;
    p.park_stride = 0;
    return ALACGPU_OK;
}

// bytes per sample the packed format can put into a slot (2 or 3; the widest stream cfg decides)
size_t packed_bytes_per_slot_int(const alacgpu_ctx* ctx) {
    size_t bps = 2;
    for (uint32_t i = 0; i < ctx->n_cfgs; i++) {
        const int ss = ctx->h_cfgs[i].ctor_sample_size ? ctx->h_cfgs[i].ctor_sample_size : ctx->h_cfgs[i].sample_size;
        bps = std::max(bps, (size_t)std::min(std::max(ss / 8, 2), 4));
    }
    return bps;
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
    case ALACGPU_ERR_COMM: return "RCCL unavailable or a collective failed";
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

int alacgpu_ctx_device(const alacgpu_ctx* ctx) { return ctx ? ctx->device : -1; }

int alacgpu_device_count(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return 0;
    int usable = 0;
    for (int d = 0; d < ndev; d++) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) usable++;
    }
    return usable;
}

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
        // (any kb SetInfo takes, AlacFile.cs:82, but 0: a value's k never exceeds 16 -- the history is bounded --, so kb > 16
        // only changes the run-length mask (1 << kb) - 1, with C#'s shift count masked to five bits)
        if (cfgs[i].rice_kmodifier < 1) return ALACGPU_ERR_UNSUPPORTED_CONFIG;
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
    if (const char* v = std::getenv("ALACGPU_DENSE")) ctx->dense = std::max(-1, std::min(std::atoi(v), 4));   // negative: auto
    if (const char* v = std::getenv("ALACGPU_HOST_CHUNKS")) ctx->host_chunks = std::max(0, std::min(std::atoi(v), N_HOST_STREAMS));
    if (const char* v = std::getenv("ALACGPU_ZERO_COPY")) ctx->zero_copy = std::atoi(v) != 0;
    int rc = ALACGPU_OK;
    do {
        if (hipSetDevice(device) != hipSuccess) { rc = ALACGPU_ERR_NO_DEVICE; break; }
        ctx->h_cfgs = (alacgpu_cfg*)std::malloc(sizeof(alacgpu_cfg) * n_cfgs);
        if (!ctx->h_cfgs) { rc = ALACGPU_ERR_NO_MEMORY; break; }
        std::memcpy(ctx->h_cfgs, cfgs, sizeof(alacgpu_cfg) * n_cfgs);
        if (hipMalloc((void**)&ctx->d_cfgs, sizeof(alacgpu_cfg_dev) * n_cfgs) != hipSuccess) { rc = ALACGPU_ERR_HIP; break; }
        if (hipMemcpy(ctx->d_cfgs, cfgs, sizeof(alacgpu_cfg) * n_cfgs, hipMemcpyHostToDevice) != hipSuccess) { rc = ALACGPU_ERR_HIP; break; }
        // (the other streams of the host-buffer pipeline, the launch slots' events and the workspace are made on first use:
        // a context per file -- the reference's AlacContext -- should cost next to nothing to open)
        if (hipStreamCreateWithFlags(&ctx->streams[0], hipStreamNonBlocking) != hipSuccess) { rc = ALACGPU_ERR_HIP; break; }
        if (!(ctx->d_cu_arrivals = cu_counters_acquire(device))) { rc = ALACGPU_ERR_HIP; break; }
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
    for (int i = 0; i < N_HOST_STREAMS; i++)
        if (ctx->streams[i]) (void)hipStreamSynchronize(ctx->streams[i]);
    for (int i = 0; i < N_SLOTS; i++) {
        launch_slot& sl = ctx->slots[i];
        if (sl.used) (void)hipEventSynchronize(sl.ev1);   // device-pointer calls on the caller's streams
        if (sl.d_flags) (void)hipFree(sl.d_flags);
        if (sl.ev0) (void)hipEventDestroy(sl.ev0);
        if (sl.ev1) (void)hipEventDestroy(sl.ev1);
    }
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->d_cu_arrivals) cu_counters_release(ctx->device);
    if (ctx->h_frame) (void)hipHostFree(ctx->h_frame);
    if (ctx->d_cfgs) (void)hipFree(ctx->d_cfgs);
    for (int i = 0; i < N_HOST_STREAMS; i++)
        if (ctx->ev_up[i]) (void)hipEventDestroy(ctx->ev_up[i]);
    for (int i = 0; i < N_HOST_STREAMS; i++)
        if (ctx->streams[i]) (void)hipStreamDestroy(ctx->streams[i]);
    if (ctx->up_stream) (void)hipStreamDestroy(ctx->up_stream);
    std::free(ctx->h_cfgs);
    delete ctx;
}

void* alacgpu_alloc_pinned(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void alacgpu_free_pinned(void* p) {
    if (p) (void)hipHostFree(p);
}

int alacgpu_decode_batch_device(alacgpu_ctx* ctx, const void* d_blob, uint64_t blob_bytes, const void* d_offsets,
                                const void* d_sizes, const void* d_cfg_idx, uint32_t n_packets, void* d_pcm_out,
                                uint32_t slot_ints, void* d_out_bytes, void* d_out_samples, void* d_status,
                                void* hip_stream) {
    if (!ctx) return ALACGPU_ERR_BAD_ARG;
    if (n_packets == 0) return ALACGPU_OK;
    alac_decode_params p;
    int rc = fill_params(ctx, p, d_blob, blob_bytes, d_offsets, d_sizes, d_cfg_idx, n_packets, d_pcm_out, slot_ints,
                         d_out_bytes, d_out_samples, d_status);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch(ctx, p, (hipStream_t)hip_stream);
}

}  // extern "C"

namespace {
// the same with channel A parked at d_park + packet * park_stride instead of inside the output slot (see alac_decode_params)
int decode_device_parked(alacgpu_ctx* ctx, const void* d_blob, uint64_t blob_bytes, const void* d_offsets, const void* d_sizes,
                         const void* d_cfg_idx, uint32_t n_packets, void* pcm_out, uint32_t slot_ints, void* d_out_bytes,
                         void* d_out_samples, void* d_status, int32_t* d_park, uint32_t park_stride, hipStream_t stream) {
    if (n_packets == 0) return ALACGPU_OK;
    alac_decode_params p;
    int rc = fill_params(ctx, p, d_blob, blob_bytes, d_offsets, d_sizes, d_cfg_idx, n_packets, pcm_out, slot_ints, d_out_bytes,
                         d_out_samples, d_status);
    if (rc) return rc;
    p.park = d_park;
    p.park_stride = park_stride;
    return launch(ctx, p, stream);
}

// pcm_out .. + bytes is page-locked host memory the device can store into: its device-side address, else null
void* device_view_of_pinned(const void* host, size_t bytes) {
    if (bytes == 0) return nullptr;
    hipPointerAttribute_t a0, a1;
    if (hipPointerGetAttributes(&a0, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipPointerGetAttributes(&a1, (const char*)host + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (a0.type != hipMemoryTypeHost || a1.type != hipMemoryTypeHost || !a0.devicePointer || !a1.devicePointer) return nullptr;
    if ((const char*)a1.devicePointer - (const char*)a0.devicePointer != (ptrdiff_t)(bytes - 1)) return nullptr;   // one mapping
    return a0.devicePointer;
}
}  // namespace

extern "C" {

#ifdef ALAC_DIAG
// Diagnostic twin of alacgpu_decode_batch_device (not part of include/alacgpu.h; tools/ only): the kernels additionally
// write 8 clock / placement stamps per workgroup into d_stamps (8 * ceil(n_packets / 8) uint64, zeroed by the caller).
int alacgpu_dbg_decode_batch_device_stamps(alacgpu_ctx* ctx, const void* d_blob, uint64_t blob_bytes, const void* d_offsets,
                                           const void* d_sizes, const void* d_cfg_idx, uint32_t n_packets, void* d_pcm_out,
                                           uint32_t slot_ints, void* d_out_bytes, void* d_out_samples, void* d_status,
                                           void* hip_stream, void* d_stamps) {
    if (!ctx || !d_stamps) return ALACGPU_ERR_BAD_ARG;
    if (n_packets == 0) return ALACGPU_OK;
    alac_decode_params p;
    int rc = fill_params(ctx, p, d_blob, blob_bytes, d_offsets, d_sizes, d_cfg_idx, n_packets, d_pcm_out, slot_ints,
                         d_out_bytes, d_out_samples, d_status);
    if (rc) return rc;
    p.dbg = (unsigned long long*)d_stamps;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch(ctx, p, (hipStream_t)hip_stream);
}
#endif

// Host buffers: the batch is cut into contiguous packet ranges (two by default, up to four), each on its own stream, so that
// the H2D copy of range k+1, the decode of range k and the D2H copy of range k-1 overlap (the two copy directions use
// different DMA engines).  Issue order: all uploads and launches first, then the downloads in range order.
int alacgpu_decode_batch(alacgpu_ctx* ctx, const uint8_t* blob, uint64_t blob_bytes, const uint64_t* offsets,
                         const uint32_t* sizes, const uint16_t* cfg_idx, uint32_t n_packets, int32_t* pcm_out,
                         uint32_t slot_ints, int32_t* out_bytes, int32_t* out_samples, int32_t* status) {
    if (!ctx) return ALACGPU_ERR_BAD_ARG;
    if (n_packets == 0) return ALACGPU_OK;
    if (!blob || !offsets || !sizes || !pcm_out || !status || slot_ints == 0) return ALACGPU_ERR_BAD_ARG;
    // measured on cfg2 (4096 packets, tools/host_path_rate.py): 1 / 2 / 4 ranges = 4.20 / 3.97 / 4.05 ms with int32 output,
    // 2.99 / 2.63 / 3.21 ms packed: the link runs at 55 GB/s either way (134 MiB of int32 PCM alone are 2.5 ms), the copies
    // from and to ordinary memory block the issuing thread, and a range's decode takes as long as the whole batch's
    int nch = ctx->host_chunks ? ctx->host_chunks : (n_packets >= 1024u ? 2 : 1);
    nch = std::min<int>(nch, (int)n_packets);
    const bool want_zc = ctx->zero_copy && device_view_of_pinned(pcm_out, sizeof(int32_t) * (size_t)n_packets * slot_ints) != nullptr;
    (void)want_zc;   // (four ranges were measured too: 3.64 / 2.61 ms against 3.56 / 2.39 with two, cfg2, page-locked buffers)
    // validate, and find the blob range every chunk needs
    uint32_t lo[N_HOST_STREAMS + 1];
    uint64_t b0[N_HOST_STREAMS], b1[N_HOST_STREAMS];
    for (int k = 0; k <= nch; k++) lo[k] = (uint32_t)(((uint64_t)n_packets * (uint64_t)k / (uint64_t)nch + 7u) & ~7ull);
    lo[0] = 0;
    lo[nch] = n_packets;
    for (int k = 1; k < nch; k++) lo[k] = std::min(lo[k], n_packets);   // (multiples of 8: whole groups of the kernel)
    uint64_t range_sum = 0;
    for (int k = 0; k < nch; k++) {
        b0[k] = blob_bytes;
        b1[k] = 0;
        for (uint32_t i = lo[k]; i < lo[k + 1]; i++) {
            if (offsets[i] > blob_bytes || (uint64_t)sizes[i] > blob_bytes - offsets[i]) return ALACGPU_ERR_BAD_ARG;
            b0[k] = std::min(b0[k], offsets[i]);
            b1[k] = std::max(b1[k], offsets[i] + sizes[i]);
        }
        if (b1[k] < b0[k]) b0[k] = b1[k] = 0;
        b0[k] &= ~(uint64_t)15;
        range_sum += b1[k] - b0[k];
    }
    if (nch > 1 && range_sum > blob_bytes + blob_bytes / 2) {   // packets not laid out in batch order: one upload
        nch = 1;
        lo[1] = n_packets;
        b0[0] = 0;
        b1[0] = blob_bytes;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // Page-locked output (alacgpu_alloc_pinned, hipHostMalloc, hipHostRegister ...): the kernels store the PCM straight into
    // the caller's memory -- the link carries it WHILE the batch decodes (50 GB/s measured: a cfg2 batch's 134 MB of int32 PCM
    // in 2.7 ms, its 67 MB of packed PCM in 1.4 ms) and there is no download behind the decode.  Channel A is then parked in
    // device memory (a read-back across the link would cost more than the decode).
    int32_t* const zc_pcm = ctx->zero_copy ? (int32_t*)device_view_of_pinned(pcm_out, sizeof(int32_t) * (size_t)n_packets * slot_ints) : nullptr;
    const uint32_t park_stride = (slot_ints + 1u) / 2u;
    // workspace carve-up (all 256-byte aligned)
    const size_t blob_sz = align_up(blob_bytes + 64, 256);
    const size_t off_sz = align_up(sizeof(uint64_t) * n_packets, 256);
    const size_t sz_sz = align_up(sizeof(uint32_t) * n_packets, 256);
    const size_t ci_sz = align_up(sizeof(uint16_t) * n_packets, 256);
    const size_t i32_sz = align_up(sizeof(int32_t) * n_packets, 256);
    const size_t pcm_sz = align_up(sizeof(int32_t) * (size_t)n_packets * (zc_pcm ? park_stride : slot_ints), 256);
    int rc = ensure_ws(ctx, blob_sz + off_sz + sz_sz + ci_sz + 3 * i32_sz + pcm_sz);
    if (rc) return rc;
    uint8_t* w = (uint8_t*)ctx->d_ws;
    uint8_t* d_blob = w; w += blob_sz;
    uint64_t* d_off = (uint64_t*)w; w += off_sz;
    uint32_t* d_sz = (uint32_t*)w; w += sz_sz;
    uint16_t* d_ci = (uint16_t*)w; w += ci_sz;
    int32_t* d_ob = (int32_t*)w; w += i32_sz;
    int32_t* d_os = (int32_t*)w; w += i32_sz;
    int32_t* d_st = (int32_t*)w; w += i32_sz;
    int32_t* d_pcm = (int32_t*)w;
    if (!ctx->up_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
    hipStream_t s0 = ctx->up_stream;
    HIP_TRY(ctx, hipMemcpyAsync(d_off, offsets, sizeof(uint64_t) * n_packets, hipMemcpyHostToDevice, s0));
    HIP_TRY(ctx, hipMemcpyAsync(d_sz, sizes, sizeof(uint32_t) * n_packets, hipMemcpyHostToDevice, s0));
    if (cfg_idx) HIP_TRY(ctx, hipMemcpyAsync(d_ci, cfg_idx, sizeof(uint16_t) * n_packets, hipMemcpyHostToDevice, s0));
    for (int k = 1; k < nch; k++)
        if (!ctx->streams[k]) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->streams[k], hipStreamNonBlocking));
    // Every upload goes through ONE stream, range after range (uploads issued on several streams run side by side and share the
    // link: all of them would finish together, at the end); range k's decode waits for its own upload only, so the first
    // range decodes -- and with page-locked output writes its PCM across the link, which is full duplex -- while the others
    // are still on their way up.
    for (int k = 0; k < nch; k++)
        if (!ctx->ev_up[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_up[k], hipEventDisableTiming));
    for (int k = 0; k < nch; k++) {
        hipStream_t s = ctx->streams[k];
        const uint32_t cnt = lo[k + 1] - lo[k];
        if (cnt == 0) continue;
        // (upload k, then launch k, then upload k + 1: a copy from ordinary memory blocks this thread while it is staged)
        if (b1[k] > b0[k]) HIP_TRY(ctx, hipMemcpyAsync(d_blob + b0[k], blob + b0[k], b1[k] - b0[k], hipMemcpyHostToDevice, s0));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_up[k], s0));
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_up[k], 0));
        if (zc_pcm)
            rc = decode_device_parked(ctx, d_blob, blob_bytes, d_off + lo[k], d_sz + lo[k], cfg_idx ? d_ci + lo[k] : nullptr, cnt,
                                      zc_pcm + (size_t)lo[k] * slot_ints, slot_ints, d_ob + lo[k], d_os + lo[k], d_st + lo[k],
                                      d_pcm + (size_t)lo[k] * park_stride, park_stride, s);
        else
            rc = alacgpu_decode_batch_device(ctx, d_blob, blob_bytes, d_off + lo[k], d_sz + lo[k], cfg_idx ? d_ci + lo[k] : nullptr,
                                             cnt, d_pcm + (size_t)lo[k] * slot_ints, slot_ints, d_ob + lo[k], d_os + lo[k],
                                             d_st + lo[k], s);
        if (rc) return rc;
    }
    const size_t pitch = sizeof(int32_t) * (size_t)slot_ints;
    const size_t packed_w = std::min(pitch, packed_bytes_per_slot_int(ctx) * (size_t)slot_ints);
    for (int k = 0; k < nch; k++) {
        hipStream_t s = ctx->streams[k];
        const uint32_t cnt = lo[k + 1] - lo[k];
        if (cnt == 0) continue;
        int32_t* dst = pcm_out + (size_t)lo[k] * slot_ints;
        const int32_t* src = d_pcm + (size_t)lo[k] * slot_ints;
        if (zc_pcm) {
            // nothing to download: the kernels wrote into the caller's memory
        } else if (ctx->out_format == ALACGPU_OUT_PACKED_LE) {
            // a slot holds at most slot_ints samples of (ctor sample size / 8) bytes: copy that much of every slot
            HIP_TRY(ctx, hipMemcpy2DAsync(dst, pitch, src, pitch, packed_w, cnt, hipMemcpyDeviceToHost, s));
        } else {
            HIP_TRY(ctx, hipMemcpyAsync(dst, src, pitch * cnt, hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(ctx, hipMemcpyAsync(status + lo[k], d_st + lo[k], sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, s));
        if (out_bytes) HIP_TRY(ctx, hipMemcpyAsync(out_bytes + lo[k], d_ob + lo[k], sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, s));
        if (out_samples) HIP_TRY(ctx, hipMemcpyAsync(out_samples + lo[k], d_os + lo[k], sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, s));
    }
    for (int k = 0; k < nch; k++)
        if (ctx->streams[k]) HIP_TRY(ctx, hipStreamSynchronize(ctx->streams[k]));
    return ALACGPU_OK;
}

// One batch on HOST buffers over several contexts -- normally one per GPU of the node -- from one process: contiguous packet
// ranges (whole groups of 8), one host thread per context, every range through alacgpu_decode_batch of its context.  The
// ranges write disjoint parts of the caller's arrays, so there is nothing to gather.
int alacgpu_decode_batch_sharded(alacgpu_ctx* const* ctxs, uint32_t n_ctxs, const uint8_t* blob, uint64_t blob_bytes,
                                 const uint64_t* offsets, const uint32_t* sizes, const uint16_t* cfg_idx, uint32_t n_packets,
                                 int32_t* pcm_out, uint32_t slot_ints, int32_t* out_bytes, int32_t* out_samples, int32_t* status) {
    if (!ctxs || n_ctxs == 0) return ALACGPU_ERR_BAD_ARG;
    for (uint32_t r = 0; r < n_ctxs; r++)
        for (uint32_t q = 0; q < r; q++)
            if (ctxs[q] == ctxs[r]) return ALACGPU_ERR_BAD_ARG;   // one context on two threads would race
    for (uint32_t r = 0; r < n_ctxs; r++)
        if (!ctxs[r] || ctxs[r]->n_cfgs != ctxs[0]->n_cfgs || ctxs[r]->out_format != ctxs[0]->out_format) return ALACGPU_ERR_BAD_ARG;
    if (n_packets == 0) return ALACGPU_OK;
    if (!blob || !offsets || !sizes || !pcm_out || !status || slot_ints == 0) return ALACGPU_ERR_BAD_ARG;
    if (n_ctxs == 1)
        return alacgpu_decode_batch(ctxs[0], blob, blob_bytes, offsets, sizes, cfg_idx, n_packets, pcm_out, slot_ints, out_bytes,
                                    out_samples, status);
    std::vector<uint32_t> lo(n_ctxs + 1);
    if (alacgpu_shard_ranges(sizes, n_packets, n_ctxs, lo.data()) != ALACGPU_OK) return ALACGPU_ERR_BAD_ARG;
    std::vector<int> rcs(n_ctxs, ALACGPU_OK);
    std::vector<std::thread> workers;
    workers.reserve(n_ctxs);
    for (uint32_t r = 0; r < n_ctxs; r++) {
        const uint32_t a = lo[r], cnt = lo[r + 1] - lo[r];
        if (cnt == 0) continue;
        workers.emplace_back([=, &rcs]() {
            rcs[r] = alacgpu_decode_batch(ctxs[r], blob, blob_bytes, offsets + a, sizes + a, cfg_idx ? cfg_idx + a : nullptr, cnt,
                                          pcm_out + (size_t)a * slot_ints, slot_ints, out_bytes ? out_bytes + a : nullptr,
                                          out_samples ? out_samples + a : nullptr, status + a);
        });
    }
    for (auto& t : workers) t.join();
    for (uint32_t r = 0; r < n_ctxs; r++)
        if (rcs[r] != ALACGPU_OK) return rcs[r];
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
    if (!ctx->h_frame) {   // pinned, sized for the widest case once (16384 samples x 2 channels)
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_frame, sizeof(int32_t) * 16384u * 2u, hipHostMallocDefault));
    }
    int32_t* pcm = ctx->h_frame;
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
        // (a one-channel element with a prediction type other than 0 carries status 3 AND the reference's output: the
        // un-predicted residuals, AlacFile.cs:484-496 with :486)
        if (st == ALACGPU_ST_OK || (st == ALACGPU_ST_UNSUPPORTED_PREDTYPE && in_bytes > 0 && (inbuffer[0] >> 5) == 0)) {
            const size_t need = (size_t)os * cfg.num_channels * (cfg.sample_size == 24 ? 3 : 1);
            if (need > out_capacity_ints) rc = ALACGPU_ERR_BAD_ARG;
            else alacgpu_expand_reference_layout(&cfg, pcm, os, outbuffer);
        }
    }
    return rc;
}

int alacgpu_set_output_format(alacgpu_ctx* ctx, int format) {
    if (!ctx || (format != ALACGPU_OUT_INT32 && format != ALACGPU_OUT_PACKED_LE)) return ALACGPU_ERR_BAD_ARG;
    ctx->out_format = (uint32_t)format;
    return ALACGPU_OK;
}

float alacgpu_last_kernel_ms(alacgpu_ctx* ctx) {
    if (!ctx || ctx->last_slot < 0) return -1.0f;
    launch_slot& sl = ctx->slots[ctx->last_slot];
    if (hipEventSynchronize(sl.ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, sl.ev0, sl.ev1) != hipSuccess) return -1.0f;
    return ms;
}

}  // extern "C"

// alacgpu_comm.hip -- multi-GPU part of the C ABI (include/alacgpu.h): the packet partition and the all-gather of decoded
// PCM over RCCL / xGMI (north_star).  One process per GPU, one alacgpu_ctx and one
// alacgpu_comm per process.  The decode itself needs no collective (packets are independent, AlacFile.cs:432-434); the
// gather is one in-place all-gather-v of fixed-stride slots, optionally overlapped with the decode range by range.
// RCCL is loaded on first use (dlopen), so a host that only decodes never depends on it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "alacgpu.h"

namespace {

// the part of rccl.h this file needs (RCCL keeps NCCL's ABI: opaque communicator, 128-byte id passed by value)
typedef struct { char internal[128]; } nccl_unique_id;
typedef void* nccl_comm_t;
enum { NCCL_SUCCESS = 0, NCCL_INT32 = 2 };
struct rccl_api {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};

rccl_api& rccl() {
    static rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        // an RCCL that is already in the process (PyTorch brings its own) is the one to use: two copies would each want the
        // device's IPC state
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names)
            if ((api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!api.handle)
            for (const char* n : names)
                if ((api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!api.handle) { api.error = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : ""); return; }
        auto sym = [&](const char* s) { void* p = dlsym(api.handle, s); if (!p) api.error = std::string("RCCL symbol missing: ") + s; return p; };
        api.GetUniqueId = (int (*)(nccl_unique_id*))sym("ncclGetUniqueId");
        api.CommInitRank = (int (*)(nccl_comm_t*, int, nccl_unique_id, int))sym("ncclCommInitRank");
        api.CommDestroy = (int (*)(nccl_comm_t))sym("ncclCommDestroy");
        api.AllGather = (int (*)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t))sym("ncclAllGather");
        api.Broadcast = (int (*)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t))sym("ncclBroadcast");
        api.Send = (int (*)(const void*, size_t, int, int, nccl_comm_t, hipStream_t))dlsym(api.handle, "ncclSend");   // (optional)
        api.Recv = (int (*)(void*, size_t, int, int, nccl_comm_t, hipStream_t))dlsym(api.handle, "ncclRecv");
        api.GroupStart = (int (*)())sym("ncclGroupStart");
        api.GroupEnd = (int (*)())sym("ncclGroupEnd");
        api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    });
    return api;
}

thread_local std::string g_comm_error;

}  // namespace

struct alacgpu_comm {
    alacgpu_ctx* ctx = nullptr;
    int device = 0, rank = 0, world = 1;
    nccl_comm_t comm = nullptr;
    hipStream_t cstream = nullptr;       // the collectives' own stream: range k gathers while range k+1 decodes
    hipEvent_t ev_ready[4] = {}, ev_done = nullptr;
    std::string last_error;
};

namespace {

#define COMM_HIP(c, expr)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (c)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
            return ALACGPU_ERR_HIP;                                                              \
        }                                                                                        \
    } while (0)
#define COMM_NCCL(c, expr)                                                                       \
    do {                                                                                         \
        int r_ = (expr);                                                                         \
        if (r_ != NCCL_SUCCESS) {                                                                \
            (c)->last_error = std::string(#expr) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r_) : "?"); \
            return ALACGPU_ERR_COMM;                                                             \
        }                                                                                        \
    } while (0)

// One in-place all-gather-v of packet ranges [first_r, first_r + count_r) of every rank r (a slot is slot_ints int32).
// Equal pieces that lie side by side in rank order are one plain in-place ncclAllGather (RCCL picks the algorithm).  Pieces
// that do not (unequal shards; the k-th piece of every shard in the overlapped form) are exchanged directly: inside one group
// every rank sends its piece to every other rank and receives theirs, in place -- on a node whose GPUs are all linked to each
// other (xGMI: 7 links per GPU) that uses every link at once, the pattern SURVEY.md section 8(e) asks for; grouped broadcasts
// (one per owner) are the fallback when the library has no ncclSend / ncclRecv.
int gather_pieces(alacgpu_comm* c, int32_t* d_full, uint32_t slot_ints, const std::vector<uint64_t>& first,
                  const std::vector<uint64_t>& count, hipStream_t s) {
    rccl_api& R = rccl();
    bool equal = true;
    for (int r = 0; r < c->world; r++) equal = equal && count[r] == count[0] && first[r] == first[0] + (uint64_t)r * count[0];
    if (equal) {   // contiguous equal pieces in rank order: the plain in-place all-gather
        if (count[0] == 0) return ALACGPU_OK;
        COMM_NCCL(c, R.AllGather(d_full + (first[0] + (uint64_t)c->rank * count[0]) * slot_ints, d_full + first[0] * slot_ints,
                                 (size_t)count[0] * slot_ints, NCCL_INT32, c->comm, s));
        return ALACGPU_OK;
    }
    if (c->world == 1) return ALACGPU_OK;   // nothing to exchange
    COMM_NCCL(c, R.GroupStart());
    if (R.Send && R.Recv) {
        const int me = c->rank;
        for (int k = 1; k < c->world; k++) {
            const int to = (me + k) % c->world, from = (me - k + c->world) % c->world;   // (a different peer pair per round)
            if (count[me]) COMM_NCCL(c, R.Send(d_full + first[me] * slot_ints, (size_t)count[me] * slot_ints, NCCL_INT32, to, c->comm, s));
            if (count[from]) COMM_NCCL(c, R.Recv(d_full + first[from] * slot_ints, (size_t)count[from] * slot_ints, NCCL_INT32, from, c->comm, s));
        }
    } else {
        for (int r = 0; r < c->world; r++) {
            if (count[r] == 0) continue;
            int32_t* piece = d_full + first[r] * slot_ints;
            COMM_NCCL(c, R.Broadcast(piece, piece, (size_t)count[r] * slot_ints, NCCL_INT32, r, c->comm, s));
        }
    }
    COMM_NCCL(c, R.GroupEnd());
    return ALACGPU_OK;
}

}  // namespace

extern "C" {

// Contiguous packet ranges for `world` ranks, cut at multiples of 8 packets (the kernels work in groups of 8).  Equal packet
// COUNTS where that leaves the ranges' summed packet bytes within 5 % of each other (equal shards gather with one plain
// all-gather); otherwise -- skewed packet sizes: SURVEY.md section 8(e) names cfg5, which mixes 1-sample packets, uncompressed
// ones and 24-bit order-30 ones -- the cuts that make the BYTES as equal as whole groups allow.
// first[r] .. first[r+1] is rank r's range; first has world + 1 entries.
int alacgpu_shard_ranges(const uint32_t* sizes, uint32_t n_packets, uint32_t world, uint32_t* first) {
    if (!first || world == 0 || (!sizes && n_packets)) return ALACGPU_ERR_BAD_ARG;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_packets; i++) total += sizes[i];
    // by count
    for (uint32_t r = 0; r <= world; r++)
        first[r] = (uint32_t)std::min<uint64_t>(n_packets, (((uint64_t)n_packets * r / world) + 7u) & ~7ull);
    first[0] = 0;
    first[world] = n_packets;
    uint64_t bmin = ~0ull, bmax = 0;
    for (uint32_t r = 0; r < world; r++) {
        uint64_t b = 0;
        for (uint32_t i = first[r]; i < first[r + 1]; i++) b += sizes[i];
        bmin = std::min(bmin, b);
        bmax = std::max(bmax, b);
    }
    if (bmax * 100 <= bmin * 105) return ALACGPU_OK;
    // by bytes
    uint64_t acc = 0;
    uint32_t i = 0;
    for (uint32_t r = 1; r < world; r++) {
        const uint64_t want = total * r / world;            // bytes that should lie before rank r
        while (i < n_packets) {                             // whole groups of 8 while that brings the cut closer to `want`
            uint64_t g = 0;
            const uint32_t e = std::min(n_packets, i + 8u);
            for (uint32_t j = i; j < e; j++) g += sizes[j];
            if (acc + g / 2 > want) break;                  // the cut before this group is the closer one
            acc += g;
            i = e;
        }
        first[r] = i;
    }
    for (uint32_t r = 1; r <= world; r++) first[r] = std::max(first[r], first[r - 1]);
    return ALACGPU_OK;
}

int alacgpu_comm_get_unique_id(void* id128) {
    if (!id128) return ALACGPU_ERR_BAD_ARG;
    rccl_api& R = rccl();
    if (!R.GetUniqueId) { g_comm_error = R.error; return ALACGPU_ERR_COMM; }
    nccl_unique_id id;
    std::memset(&id, 0, sizeof(id));
    const int r = R.GetUniqueId(&id);
    if (r != NCCL_SUCCESS) { g_comm_error = std::string("ncclGetUniqueId: ") + R.GetErrorString(r); return ALACGPU_ERR_COMM; }
    std::memcpy(id128, &id, sizeof(id));
    return ALACGPU_OK;
}

int alacgpu_comm_create(alacgpu_ctx* ctx, const void* id128, int rank, int world, alacgpu_comm** out) {
    if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) return ALACGPU_ERR_BAD_ARG;
    *out = nullptr;
    rccl_api& R = rccl();
    if (!R.CommInitRank || !R.AllGather || !R.Broadcast || !R.GroupStart || !R.GroupEnd || !R.CommDestroy) {
        g_comm_error = R.error;
        return ALACGPU_ERR_COMM;
    }
    alacgpu_comm* c = new (std::nothrow) alacgpu_comm();
    if (!c) return ALACGPU_ERR_NO_MEMORY;
    c->ctx = ctx;
    c->device = alacgpu_ctx_device(ctx);
    c->rank = rank;
    c->world = world;
    int rc = ALACGPU_OK;
    do {
        if (hipSetDevice(c->device) != hipSuccess) { rc = ALACGPU_ERR_NO_DEVICE; break; }
        nccl_unique_id id;
        std::memcpy(&id, id128, sizeof(id));
        const int r = R.CommInitRank(&c->comm, world, id, rank);
        if (r != NCCL_SUCCESS) { g_comm_error = std::string("ncclCommInitRank: ") + R.GetErrorString(r); rc = ALACGPU_ERR_COMM; break; }
        if (hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking) != hipSuccess) { rc = ALACGPU_ERR_HIP; break; }
        for (auto& e : c->ev_ready)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { rc = ALACGPU_ERR_HIP; break; }
        if (rc == ALACGPU_OK && hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) rc = ALACGPU_ERR_HIP;
    } while (0);
    if (rc != ALACGPU_OK) {
        alacgpu_comm_destroy(c);
        return rc;
    }
    *out = c;
    return ALACGPU_OK;
}

void alacgpu_comm_destroy(alacgpu_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->cstream) (void)hipStreamSynchronize(c->cstream);
    if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
    for (auto& e : c->ev_ready)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->cstream) (void)hipStreamDestroy(c->cstream);
    delete c;
}

int alacgpu_comm_rank(const alacgpu_comm* c) { return c ? c->rank : -1; }
int alacgpu_comm_world(const alacgpu_comm* c) { return c ? c->world : 0; }
const char* alacgpu_comm_last_error(const alacgpu_comm* c) { return c ? c->last_error.c_str() : g_comm_error.c_str(); }

// d_full: the whole batch's PCM slots in global packet order on THIS rank's GPU; rank r's packets are
// first[r] .. first[r+1] and this rank's own range is already decoded in place.  Asynchronous on hip_stream.
int alacgpu_allgather_pcm(alacgpu_comm* c, void* d_full, const uint32_t* first, uint32_t slot_ints, void* hip_stream) {
    if (!c || !d_full || !first || slot_ints == 0) return ALACGPU_ERR_BAD_ARG;
    COMM_HIP(c, hipSetDevice(c->device));
    std::vector<uint64_t> f(c->world), n(c->world);
    for (int r = 0; r < c->world; r++) {
        if (first[r + 1] < first[r]) return ALACGPU_ERR_BAD_ARG;
        f[r] = first[r];
        n[r] = first[r + 1] - first[r];
    }
    return gather_pieces(c, (int32_t*)d_full, slot_ints, f, n, (hipStream_t)hip_stream);
}

// Decode this rank's range of a batch whose packets are resident in HBM and all-gather the PCM, overlapped: the range is
// decoded in n_chunks pieces on hip_stream; piece k's gather runs on the communicator's own stream while piece k+1
// decodes.  When the call returns everything is enqueued; hip_stream has passed the last gather when it gets there.
// d_offsets / d_sizes / d_cfg_idx / d_out_bytes / d_out_samples / d_status / d_full_pcm are indexed by GLOBAL packet number.
int alacgpu_decode_allgather_device(alacgpu_ctx* ctx, alacgpu_comm* c, const void* d_blob, uint64_t blob_bytes,
                                    const void* d_offsets, const void* d_sizes, const void* d_cfg_idx, const uint32_t* first,
                                    void* d_full_pcm, uint32_t slot_ints, void* d_out_bytes, void* d_out_samples,
                                    void* d_status, uint32_t n_chunks, void* hip_stream) {
    if (!ctx || !c || c->ctx != ctx || !first || !d_full_pcm || slot_ints == 0) return ALACGPU_ERR_BAD_ARG;
    n_chunks = std::max(1u, std::min(n_chunks, 4u));
    COMM_HIP(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)hip_stream;
    const int W = c->world;
    // piece k of rank r: whole groups of 8 packets
    auto cut = [&](int r, uint32_t k) {
        const uint64_t lo = first[r], cnt = first[r + 1] - first[r];
        if (k >= n_chunks) return lo + cnt;
        return lo + std::min<uint64_t>(cnt, ((cnt * k / n_chunks) + 7u) & ~7ull);
    };
    for (int r = 0; r < W; r++)
        if (first[r + 1] < first[r]) return ALACGPU_ERR_BAD_ARG;
    for (uint32_t k = 0; k < n_chunks; k++) {
        const uint64_t a = cut(c->rank, k), b = cut(c->rank, k + 1);
        if (b > a) {
            const int rc = alacgpu_decode_batch_device(
                ctx, d_blob, blob_bytes, (const uint64_t*)d_offsets + a, (const uint32_t*)d_sizes + a,
                d_cfg_idx ? (const void*)((const uint16_t*)d_cfg_idx + a) : nullptr, (uint32_t)(b - a),
                (int32_t*)d_full_pcm + a * slot_ints, slot_ints, d_out_bytes ? (void*)((int32_t*)d_out_bytes + a) : nullptr,
                d_out_samples ? (void*)((int32_t*)d_out_samples + a) : nullptr, (int32_t*)d_status + a, s);
            if (rc != ALACGPU_OK) { c->last_error = alacgpu_last_error(ctx); return rc; }
        }
        COMM_HIP(c, hipEventRecord(c->ev_ready[k], s));
        COMM_HIP(c, hipStreamWaitEvent(c->cstream, c->ev_ready[k], 0));
        std::vector<uint64_t> f(W), n(W);
        for (int r = 0; r < W; r++) {
            f[r] = cut(r, k);
            n[r] = cut(r, k + 1) - f[r];
        }
        const int rc = gather_pieces(c, (int32_t*)d_full_pcm, slot_ints, f, n, c->cstream);
        if (rc != ALACGPU_OK) return rc;
    }
    COMM_HIP(c, hipEventRecord(c->ev_done, c->cstream));
    COMM_HIP(c, hipStreamWaitEvent(s, c->ev_done, 0));
    return ALACGPU_OK;
}

}  // extern "C"

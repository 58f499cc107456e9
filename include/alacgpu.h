/*
 * alacgpu.h -- C ABI of libalacgpu.so: the MI355X (gfx950) batched ALAC frame-decode path.
 *
 * This is the drop-in boundary for teekay/ALAC.NET's per-packet decode seam.  The reference has
 * no FFI; the seam is the managed call
 *     AlacContext.cs:54-55   _alac = new AlacFile(SampleSize, NumChannels); _alac.SetInfo(CodecData);
 *     AlacContext.cs:197     var outputBytes = _alac.DecodeFrame(_readBuffer, pDestBuffer);
 * and each entry point below names the reference member it replaces.  A C# host binds these with
 * [DllImport("alacgpu")].  Plain pointers and sizes only; caller allocates
 * everything; the library never keeps a caller pointer past the call; one ctx per host thread
 * (a ctx is not thread-safe; one thread may keep up to 8 asynchronous device-pointer calls in flight on
 * any streams, see alacgpu_decode_batch_device).  One ctx drives one GPU: a multi-GPU host creates one ctx
 * per device (alacgpu_device_count) -- in this repository one process per GPU (bench.py, sharding.py).
 *
 * Return codes: 0 = the batch ran (inspect status[]), < 0 = batch-level failure.
 * There is NO CPU fallback: if no gfx950 device / kernel image is usable, create fails.
 */
#ifndef ALACGPU_H
#define ALACGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALACGPU_VERSION 3

/* Stream configuration = the AlacFile ctor args + what AlacFile.SetInfo keeps
 * (AlacFile.cs:16-20 and :63-93; CodecData byte offsets in brackets). 12 bytes, blittable. */
typedef struct {
    uint32_t max_samples_per_frame; /* [24..27] BE32, AlacFile.cs:72  */
    uint8_t  sample_size;           /* [29]            AlacFile.cs:76  (16 or 24 decode; others -> status) */
    uint8_t  rice_history_mult;     /* [30]            AlacFile.cs:78  */
    uint8_t  rice_initial_history;  /* [31]            AlacFile.cs:80  */
    uint8_t  rice_kmodifier;        /* [32]            AlacFile.cs:82  (1..255; 0 is refused) */
    uint8_t  num_channels;          /* ctor arg,       AlacFile.cs:18  (1 or 2) */
    uint8_t  ctor_sample_size;      /* ctor arg samplesize (AlacFile.cs:19); 0 = same as sample_size */
    uint8_t  reserved;
} alacgpu_cfg;

/* Per-packet status[] values.  The reference signals these by exceptions / silent no-ops. */
enum {
    ALACGPU_ST_OK = 0,
    ALACGPU_ST_UNSUPPORTED_ELEMENT = 1,     /* channels field not 0/1: reference decodes nothing (AlacFile.cs:437,:577) */
    ALACGPU_ST_UNSUPPORTED_SAMPLE_SIZE = 2, /* Exception("FIXME: unimplemented sample size N") (AlacFile.cs:574,:715) */
    ALACGPU_ST_UNSUPPORTED_PREDTYPE = 3,    /* Exception("FIXME: unhandled predicition type: N") (AlacFile.cs:650,:660) */
    ALACGPU_ST_BAD_SAMPLE_COUNT = 4,        /* hassize count <= 0, > 16384 or > slot (IndexOutOfRangeException) */
    ALACGPU_ST_OVERRUN = 5,                 /* bitstream ran past the packet / zero run past the scratch (AlacFile.cs:242) */
    ALACGPU_ST_REF_THROWS = 6,              /* N == 0 && n > 4096: Array.Copy ArgumentException (AlacFile.cs:264-265) */
    ALACGPU_ST_UNSUPPORTED_PARAMS = 7       /* header/parameter combination outside the supported domain */
};

/* Batch-level return codes */
enum {
    ALACGPU_OK = 0,
    ALACGPU_ERR_BAD_ARG = -1,
    ALACGPU_ERR_NO_DEVICE = -2,          /* no usable gfx950 GPU: there is no CPU fallback */
    ALACGPU_ERR_HIP = -3,                /* a HIP runtime call failed; see alacgpu_last_error */
    ALACGPU_ERR_UNSUPPORTED_CONFIG = -4, /* a cfg is outside the kernel's domain (rice_kmodifier 0, channels not 1/2) */
    ALACGPU_ERR_NO_MEMORY = -5,
    ALACGPU_ERR_COMM = -6                /* RCCL could not be loaded or a collective failed; see alacgpu_comm_last_error */
};

typedef struct alacgpu_ctx alacgpu_ctx;

int alacgpu_version(void);

/* Number of usable (gfx950) devices; 0 when there is none (then alacgpu_create fails: no CPU fallback). */
int alacgpu_device_count(void);

/* Replaces `new AlacFile(samplesize, numchannels)` + `SetInfo(codecData)` (AlacContext.cs:54-55)
 * for one or more streams at once (a batch may mix streams through cfg_idx[]).
 * device = HIP device ordinal. */
int alacgpu_create(const alacgpu_cfg* cfgs, uint32_t n_cfgs, int device, alacgpu_ctx** out_ctx);

void alacgpu_destroy(alacgpu_ctx* ctx);

/* Parses the int-per-byte CodecData array exactly as AlacFile.SetInfo does (AlacFile.cs:63-93). */
int alacgpu_cfg_from_codec_data(const int32_t* codec_data_ints, uint32_t n_ints, int samplesize, int numchannels,
                                alacgpu_cfg* out_cfg);

/*
 * Batched AlacFile.DecodeFrame (AlacFile.cs:428-719) on HOST buffers: H2D, decode kernel, D2H; blocking.
 * Batches of 1024 packets and more are cut into two contiguous packet ranges on separate streams so that the
 * upload of one range, the decode of the previous and the download of the one before overlap (needs the packets to
 * lie in the blob in batch order; any other layout works too, with one upload).
 * Bytes outside [offsets[p], offsets[p]+sizes[p]) are never interpreted as part of packet p: a packet that is cut
 * short decodes as if zero bits followed (and reports ALACGPU_ST_OVERRUN).
 *   blob/blob_bytes      concatenated raw ALAC packets
 *   offsets[p], sizes[p] packet p = blob[offsets[p] .. offsets[p]+sizes[p])   (any byte alignment)
 *   cfg_idx[p]           stream cfg of packet p; NULL = all 0
 *   pcm_out              packet p decodes to pcm_out + p*slot_ints, ONE int32 PER SAMPLE interleaved by the
 *                        stream's num_channels (16-bit: exactly the ints DecodeFrame stores; 24-bit: the
 *                        sample sign-extended -- DecodeFrame's byte-per-int layout is alacgpu_expand_reference_layout)
 *   slot_ints            >= max n*num_channels over the batch.  What a slot holds beyond the packet's own output
 *                        (n*num_channels ints, or out_bytes[p] bytes in the packed format) is unspecified: the kernels
 *                        use the slot as scratch while they decode
 *   out_bytes[p]         DecodeFrame's return value (AlacFile.cs:718); may be NULL
 *   out_samples[p]       samples per channel in packet p; may be NULL
 *   status[p]            ALACGPU_ST_*
 */
int alacgpu_decode_batch(alacgpu_ctx* ctx, const uint8_t* blob, uint64_t blob_bytes, const uint64_t* offsets,
                         const uint32_t* sizes, const uint16_t* cfg_idx, uint32_t n_packets, int32_t* pcm_out,
                         uint32_t slot_ints, int32_t* out_bytes, int32_t* out_samples, int32_t* status);

/*
 * The same batch over SEVERAL contexts from one process -- normally one context per GPU of the node (alacgpu_device_count,
 * alacgpu_create with device = 0, 1, ...): the packets are cut into n_ctxs contiguous ranges (alacgpu_shard_ranges: whole
 * groups of 8 packets, balanced by packet bytes), one host thread per context runs alacgpu_decode_batch on its range, and
 * every range writes its own part of the caller's arrays (nothing to gather).  All contexts must have been created with
 * the same cfgs and output format and must be DISTINCT (a context is not thread-safe: ALACGPU_ERR_BAD_ARG otherwise).
 * This is how a single-process host (the C# AlacContext) uses every GPU of a node with results in HOST memory; for
 * results that stay in HBM on every GPU see alacgpu_comm_* below (one process per GPU).
 */
int alacgpu_decode_batch_sharded(alacgpu_ctx* const* ctxs, uint32_t n_ctxs, const uint8_t* blob, uint64_t blob_bytes,
                                 const uint64_t* offsets, const uint32_t* sizes, const uint16_t* cfg_idx, uint32_t n_packets,
                                 int32_t* pcm_out, uint32_t slot_ints, int32_t* out_bytes, int32_t* out_samples,
                                 int32_t* status);

/*
 * The packet partition every multi-GPU entry point uses: contiguous ranges, cut at multiples of
 * 8 packets, whose summed packet bytes are as equal as such cuts allow.  first[] gets world + 1 entries; rank r owns
 * packets first[r] .. first[r+1].  Pure host arithmetic (no GPU needed).
 */
int alacgpu_shard_ranges(const uint32_t* sizes, uint32_t n_packets, uint32_t world, uint32_t* first);

/*
 * Multi-GPU, one process per GPU (north_star: "packet batches shard trivially across the 8 GPUs of one node with an RCCL
 * all-gather of decoded PCM over xGMI"; there is no counterpart in the reference, whose AlacContext.cs:195-197 decodes one
 * packet at a time on the host).  Every process makes its alacgpu_ctx, rank 0 makes an id (alacgpu_comm_get_unique_id:
 * 128 bytes) and hands it to the others by whatever channel the host has, and all call alacgpu_comm_create (collective:
 * ncclCommInitRank).  RCCL is loaded when the first of these functions is called (librccl.so, the copy already in the
 * process if there is one); a host that never calls them does not need it.
 *   alacgpu_allgather_pcm          d_full holds the WHOLE batch's slots in global packet order on this rank's GPU; this
 *                                  rank's packets first[rank] .. first[rank+1] are decoded in place; on return (asynchronous
 *                                  on hip_stream) every rank holds every packet: one in-place all-gather(-v) of int32
 *   alacgpu_decode_allgather_device  decode + gather overlapped: this rank's range is decoded in n_chunks (1..4) pieces on
 *                                  hip_stream, piece k is gathered on the communicator's own stream while piece k+1
 *                                  decodes; hip_stream waits for the last gather.  All device arrays are indexed by
 *                                  GLOBAL packet number (the batch's metadata is resident on every GPU).
 * All ranks must make the same calls with the same first[] / n_chunks.  Return codes as elsewhere; ALACGPU_ERR_COMM =
 * RCCL missing or a collective failed (alacgpu_comm_last_error(comm), or (NULL) for the calling thread's last failure).
 */
#define ALACGPU_COMM_ID_BYTES 128
typedef struct alacgpu_comm alacgpu_comm;
int alacgpu_comm_get_unique_id(void* id128);
int alacgpu_comm_create(alacgpu_ctx* ctx, const void* id128, int rank, int world, alacgpu_comm** out_comm);
void alacgpu_comm_destroy(alacgpu_comm* comm);
int alacgpu_comm_rank(const alacgpu_comm* comm);
int alacgpu_comm_world(const alacgpu_comm* comm);
const char* alacgpu_comm_last_error(const alacgpu_comm* comm);
int alacgpu_allgather_pcm(alacgpu_comm* comm, void* d_full_pcm, const uint32_t* first, uint32_t slot_ints, void* hip_stream);
int alacgpu_decode_allgather_device(alacgpu_ctx* ctx, alacgpu_comm* comm, const void* d_blob, uint64_t blob_bytes,
                                    const void* d_offsets, const void* d_sizes, const void* d_cfg_idx, const uint32_t* first,
                                    void* d_full_pcm, uint32_t slot_ints, void* d_out_bytes, void* d_out_samples,
                                    void* d_status, uint32_t n_chunks, void* hip_stream);
/* the device ordinal a context was created on */
int alacgpu_ctx_device(const alacgpu_ctx* ctx);

/*
 * Same, on DEVICE buffers already resident in HBM (all pointers are device pointers), asynchronous on
 * `hip_stream` (a hipStream_t; NULL = default stream).  d_blob must be 16-byte aligned and readable up to
 * blob_bytes rounded up to 16 (ALACGPU_ERR_BAD_ARG otherwise; the other arrays need their natural alignment).
 * Outputs as above; d_out_bytes / d_out_samples may be NULL.  Up to 8 calls may be in flight at once on one ctx, on
 * the same or on different streams (each owns its scratch until it has finished; a ninth call waits for the oldest).
 * The caller keeps every buffer alive and unchanged until the stream has passed the call.
 * Throughput: a launch of a few thousand packets is bound by the length of one packet's serial chain, not by the chip; a
 * caller with a stream of such batches keeps TWO in flight on two streams (0.70 -> about 0.5 ms per batch of 4096 packets,
 * bench.py key two_in_flight; streams that share one of the runtime's hardware queues do not overlap) -- or makes its batches
 * bigger.
 */
int alacgpu_decode_batch_device(alacgpu_ctx* ctx, const void* d_blob, uint64_t blob_bytes, const void* d_offsets,
                                const void* d_sizes, const void* d_cfg_idx, uint32_t n_packets, void* d_pcm_out,
                                uint32_t slot_ints, void* d_out_bytes, void* d_out_samples, void* d_status,
                                void* hip_stream);

/* Single-packet drop-in for `int DecodeFrame(byte[] inbuffer, int[] outbuffer)` (AlacFile.cs:428):
 * writes the reference's own int[] layout (24-bit: one int per byte) and returns its byte count in
 * *out_bytes.  status as above (the C# shim rethrows the reference's exceptions from it). */
int alacgpu_decode_frame(alacgpu_ctx* ctx, uint32_t cfg_index, const uint8_t* inbuffer, uint32_t in_bytes,
                         int32_t* outbuffer, uint32_t out_capacity_ints, int32_t* out_bytes, int32_t* status);

/* Host-side reshape: canonical int32-per-sample -> the exact int[] DecodeFrame writes
 * (AlacFile.cs:390-395,:555-557 for 24-bit; identity for 16-bit).  Returns ints written. */
size_t alacgpu_expand_reference_layout(const alacgpu_cfg* cfg, const int32_t* pcm, int32_t n_samples,
                                       int32_t* ref_ints);

/* AlacContext.FormatSamples (AlacContext.cs:214-256): reference int[] -> little-endian PCM bytes. */
size_t alacgpu_format_samples(int bytes_per_sample, const int32_t* ref_ints, int32_t count_bytes, uint8_t* dst);

/* Kernel time of the most recent launch of this ctx, from HIP events recorded on the launch stream around it
 * (milliseconds; < 0 if unavailable).  Waits for that launch.  After a host-buffer call that was cut into ranges
 * this is the last range's launch. */
float alacgpu_last_kernel_ms(alacgpu_ctx* ctx);

/* Output layout of the batch entry points.  ALACGPU_OUT_INT32 (default): one int32 per sample, as documented
 * above.  ALACGPU_OUT_PACKED_LE: the bytes AlacContext.Read hands out -- AlacContext.FormatSamples
 * (AlacContext.cs:214-256) fused into the kernel's store: packet p's little-endian PCM (2 or 3 bytes per sample,
 * interleaved) starts at (uint8_t*)(pcm_out + p*slot_ints) and is out_bytes[p] long.  The slot stride is unchanged;
 * alacgpu_decode_batch then copies back only the part of each slot the widest stream cfg can fill (2 or 3 bytes per
 * slot int), the rest of the caller's slot is left untouched. */
enum { ALACGPU_OUT_INT32 = 0, ALACGPU_OUT_PACKED_LE = 1 };
int alacgpu_set_output_format(alacgpu_ctx* ctx, int format);

/* Page-locked host memory for batch buffers (blob, offsets, pcm_out ...).  Optional -- every entry point takes ordinary memory
 * too -- but the faster choice: when pcm_out of alacgpu_decode_batch / alacgpu_decode_frame is page-locked (from here, from
 * hipHostMalloc, or registered with hipHostRegister) the kernels store the PCM straight into it while the batch decodes and
 * there is no download behind the decode (cfg2, 4096 packets: 3.55 / 2.39 ms int32 / packed against 3.70 / 2.50 ms from
 * ordinary memory); what a slot holds beyond the packet's own output is then left untouched.
 * ALACGPU_ZERO_COPY=0 in the environment keeps the copying path (A/B).  NULL on failure. */
void* alacgpu_alloc_pinned(size_t bytes);
void alacgpu_free_pinned(void* p);

const char* alacgpu_strerror(int rc);
const char* alacgpu_status_string(int status);
const char* alacgpu_last_error(alacgpu_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif

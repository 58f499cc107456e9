/*
 * alac_oracle.h -- CPU ORACLE for the ALAC frame-decode path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a scalar C restatement of teekay/ALAC.NET's ALACDecoder/AlacFile.cs (the reference
 * cannot be compiled or run in this image: it is C# and there is no .NET runtime).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the
 * checker / reported CPU baseline -- never as the product path.
 *
 * PARITY PINNING: the reference ships no tests, golden vectors or audio fixtures (SURVEY.md
 * section 4 / 8c), so this oracle is "parity unpinned" by reference fixtures.  It is pinned
 * instead by the hand-derived known-answer tests of SURVEY.md App. C (tests/test_oracle_kat.py),
 * each derived line by line from AlacFile.cs, and by an independent encoder round trip.
 *
 * All arithmetic follows C# `int` semantics: 32-bit two's complement, wrapping add/sub/mul,
 * arithmetic >> on int, shift counts masked to 5 bits, '/' truncating toward zero.
 */
#ifndef ALAC_ORACLE_H
#define ALAC_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors the AlacFile ctor args + the fields SetInfo() keeps (AlacFile.cs:16-20, :63-93). */
typedef struct {
    uint32_t max_samples_per_frame; /* CodecData[24..27], AlacFile.cs:72 */
    uint8_t  sample_size;           /* CodecData[29],     AlacFile.cs:76 */
    uint8_t  rice_history_mult;     /* CodecData[30],     AlacFile.cs:78 */
    uint8_t  rice_initial_history;  /* CodecData[31],     AlacFile.cs:80 */
    uint8_t  rice_kmodifier;        /* CodecData[32],     AlacFile.cs:82 */
    uint8_t  num_channels;          /* ctor arg,          AlacFile.cs:18 */
    uint8_t  ctor_sample_size;      /* ctor arg `samplesize` (-> _bytespersample), AlacFile.cs:19;
                                       0 means "same as sample_size" */
    uint8_t  reserved;
} alac_oracle_cfg;

/* Per-packet status codes.  Same numbering as include/alacgpu.h (ALACGPU_ST_*). */
enum {
    ALAC_ORACLE_OK = 0,
    ALAC_ORACLE_UNSUPPORTED_ELEMENT = 1,     /* channels field not 0/1: reference decodes nothing (AlacFile.cs:437,:577) */
    ALAC_ORACLE_UNSUPPORTED_SAMPLE_SIZE = 2, /* reference throws "FIXME: unimplemented sample size" (:574,:715) or no-ops */
    ALAC_ORACLE_UNSUPPORTED_PREDTYPE = 3,    /* stereo: throws (:650,:660); mono: stale data (:488-496) */
    ALAC_ORACLE_BAD_SAMPLE_COUNT = 4,        /* hassize count <= 0 or > slot capacity (reference: no-op / IndexOutOfRange) */
    ALAC_ORACLE_OVERRUN = 5,                 /* bitstream ran past the packet, or zero-run past the frame (:240-244) */
    ALAC_ORACLE_REF_THROWS = 6,              /* N==0 && n>4096: Array.Copy throws ArgumentException (:264-265) */
    ALAC_ORACLE_UNSUPPORTED_PARAMS = 7       /* ub/rss combination outside the supported domain */
};

/* Parse the 48-int (one int per byte) CodecData array exactly as AlacFile.SetInfo does
 * (AlacFile.cs:63-93).  num_channels / ctor_sample_size are the AlacFile ctor arguments. */
void alac_oracle_set_info(const int32_t* codec_data_ints, int samplesize, int numchannels,
                          alac_oracle_cfg* out_cfg);

/*
 * DecodeFrame restatement (AlacFile.cs:428-719) with the output in the build's canonical
 * layout: ONE int32 PER SAMPLE, interleaved by the file-level channel count:
 *     pcm[i * num_channels + c]
 * 16-bit streams: exactly the ints the reference stores (full 32-bit, not truncated, :353-354).
 * 24-bit streams: the sample sign-extended from 24 bits; the reference's byte-per-int layout
 * (:390-395) is `alac_oracle_expand_reference_layout` of it.
 *   packet/packet_size : the raw ALAC packet (reads past the end return 0 and raise OVERRUN
 *                        when consumed bits exceed the packet)
 *   pcm/pcm_capacity   : output slot, in ints
 *   out_bytes          : DecodeFrame's return value (outputsamples * _bytespersample, :718)
 *   out_samples        : samples per channel decoded (outputsamples)
 * Returns the per-packet status.
 */
int alac_oracle_decode_frame(const alac_oracle_cfg* cfg, const uint8_t* packet, size_t packet_size,
                             int32_t* pcm, size_t pcm_capacity, int32_t* out_bytes,
                             int32_t* out_samples);

/* Expand canonical int32-per-sample PCM to the exact int[] the reference's DecodeFrame writes
 * (16-bit: identity copy; 24-bit: three byte-valued ints per sample, AlacFile.cs:390-395,:555-557).
 * Returns the number of ints written. */
size_t alac_oracle_expand_reference_layout(const alac_oracle_cfg* cfg, const int32_t* pcm,
                                           int32_t n_samples, int32_t* ref_ints);

/* AlacContext.FormatSamples restatement (AlacContext.cs:214-256) on the REFERENCE int[] layout.
 * `count_bytes` is the byte count DecodeFrame returned.  Returns bytes written. */
size_t alac_oracle_format_samples(int bps, const int32_t* ref_ints, int32_t count_bytes, uint8_t* dst);

/* Batch helper with the same argument meaning as alacgpu_decode_batch (include/alacgpu.h);
 * used as the checker and as bench.py's cpu_baseline ("port").  n_threads<=1 => scalar loop. */
int alac_oracle_decode_batch(const alac_oracle_cfg* cfgs, uint32_t n_cfgs, const uint8_t* blob,
                             const uint64_t* offsets, const uint32_t* sizes, const uint16_t* cfg_idx,
                             uint32_t n_packets, int32_t* pcm_out, uint32_t slot_ints,
                             int32_t* out_bytes, int32_t* out_samples, int32_t* status, int n_threads);

/* Unit-test hooks for the inner functions (KATs of SURVEY.md App. C). */
int  alac_oracle_count_leading_zeros(int32_t x);                       /* AlacFile.cs:170-191 */
void alac_oracle_predictor(int32_t* buf, int n, int rss, int32_t* coef, int ncoef, int q); /* :256-336 */
void alac_oracle_deinterlace16(const int32_t* a, const int32_t* b, int32_t* out, int nc, int n,
                               int shift, int weight);                 /* :338-367 */
int  alac_oracle_rice_decode(const uint8_t* bits, size_t nbytes, int32_t* out, int n, int rss,
                             int init_hist, int kmod, int hist_mult, int* end_bitpos); /* :214-252 */

#ifdef __cplusplus
}
#endif
#endif

/*
 * alac_synth.h -- synthetic ALAC packet generator (an ALAC *encoder*), used by tests/ and
 * bench.py to make valid packets.  The reference ships no audio, no encoder and no fixtures
 * (SURVEY.md section 4), so inputs have to be made here.
 *
 * It is the encoder-side inverse of the decode path: the forward adaptive
 * predictor and the adaptive Golomb-Rice writer run the *decoder's* state machines
 * (AlacFile.cs:214-252 history/k/signModifier; :256-336 coefficient adaptation) so that
 * decode(encode(pcm)) == pcm under the reference's semantics.  Not part of the product path and
 * not part of the oracle: it shares no code with either.
 */
#ifndef ALAC_SYNTH_H
#define ALAC_SYNTH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* One packet's encoding recipe.  Plain-old-data; mirrored by a numpy structured dtype. */
typedef struct {
    uint32_t n;                /* samples per channel in this packet */
    uint32_t max_samples_per_frame; /* stream config; hassize is set when n differs or force_hassize */
    uint8_t sample_size;       /* 16 | 24 */
    uint8_t stereo;            /* 0 mono element, 1 stereo element */
    uint8_t ub;                /* uncompressedBytes ("bytes shifted") 0..2 */
    uint8_t escape;            /* 1 = isnotcompressed packet */
    uint8_t force_hassize;
    uint8_t pred_order[2];     /* N per channel 0..31 */
    uint8_t quant[2];          /* predictionQuantitization 0..15 */
    uint8_t ricemod[2];        /* ricemodifier 0..7 */
    uint8_t pred_type[2];      /* 0 normally; non-zero only for negative tests */
    uint8_t mix_shift;         /* interlacingShift */
    uint8_t mix_weight;        /* interlacingLeftweight 0..255 */
    uint8_t coef_mode;         /* 0 = Levinson-Durbin on the packet, 1 = coefs[] given, 2 = zeros */
    uint8_t rice_history_mult, rice_initial_history, rice_kmodifier; /* stream config (pb, mb, kb) */
    int8_t  channels_field;    /* -1 = auto (0 mono / 1 stereo); else raw 3-bit value (negative tests) */
    uint8_t pad[4];
    int16_t coefs[2][32];
} alac_synth_pkt;

/* Signal recipe for the built-in PCM source. */
typedef struct {
    uint64_t seed;             /* per-packet seed = seed + packet_index */
    float    amp_lo_log2, amp_hi_log2; /* sinusoid amplitude range, in bits, relative to 16-bit full scale */
    float    noise_sigma;      /* Gaussian noise sigma at 16-bit scale */
    float    silence_prob;     /* fraction of packets with a digital-silence segment */
    uint32_t silence_min, silence_max;
    float    lr_corr;          /* R = lr_corr * L + own noise */
} alac_synth_signal;

/* Encode one packet from interleaved PCM (pcm[i*ch + c], ch = 1 or 2 per desc->stereo).
 * Returns the packet size in bytes, or 0 when `cap` is too small / parameters are invalid. */
size_t alac_synth_encode_packet(const alac_synth_pkt* desc, const int32_t* pcm, uint8_t* out, size_t cap);

/* Generate the PCM for packet `index` (interleaved, ch channels) */
void alac_synth_make_pcm(const alac_synth_signal* sig, uint64_t index, int sample_size, int ch, uint32_t n,
                         int32_t* pcm);

/* Generate + encode a whole batch.  descs[n_packets]; blob receives the packets back to back
 * (each start 16-byte aligned); offsets/sizes per packet.  If pcm_out != NULL the source PCM of packet p
 * is stored at pcm_out + p*pcm_slot_ints (interleaved) for round-trip checks.
 * Returns total blob bytes used, or 0 on overflow. */
size_t alac_synth_make_batch(const alac_synth_pkt* descs, uint32_t n_packets, const alac_synth_signal* sig,
                             uint64_t first_index, int n_threads, uint8_t* blob, size_t blob_cap,
                             uint64_t* offsets, uint32_t* sizes, int32_t* pcm_out, uint32_t pcm_slot_ints);

/* Upper bound of one packet's encoded size for sizing the blob. */
size_t alac_synth_max_packet_bytes(uint32_t n, int sample_size, int stereo);

#ifdef __cplusplus
}
#endif
#endif

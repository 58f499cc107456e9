/*
 * alac_synth.c -- synthetic ALAC packet generator (encoder).  See alac_synth.h.
 *
 * Encoder-side inverse of the decode path, written from the bitstream layout the reference's
 * decoder parses (AlacFile.cs:428-719).  The state machines that must agree
 * with the decoder bit for bit are annotated with the decoder lines they mirror.
 */
#include "alac_synth.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- MSB-first bit writer ------------------------------------------------------------------ */
typedef struct {
    uint8_t* buf;
    size_t cap;
    size_t bitpos;
    int overflow;
} bitwr;

static void put_bits(bitwr* w, uint32_t value, int nbits) {
    for (int i = nbits - 1; i >= 0; i--) {
        size_t byte = w->bitpos >> 3;
        if (byte >= w->cap) { w->overflow = 1; return; }
        int bit = (value >> i) & 1;
        int sh = 7 - (int)(w->bitpos & 7);
        w->buf[byte] = (uint8_t)((w->buf[byte] & ~(1u << sh)) | ((uint32_t)bit << sh));
        w->bitpos++;
    }
}

/* ---- wrapping int helpers ------------------------------------------------------------------- */
static inline int32_t w_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t w_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t w_mul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t w_shl(int32_t a, int n) { return (int32_t)((uint32_t)a << (n & 31)); }
static inline int32_t w_sar(int32_t a, int n) { return a >> (n & 31); }
static inline int32_t sx(int32_t v, int rss) { int m = 32 - rss; return w_sar(w_shl(v, m), m); }

/* clz with the decoder's convention for 0 (AlacFile.cs:170-191 returns 40) */
static inline int clz_q(int32_t x) { return x == 0 ? 40 : __builtin_clz((uint32_t)x); }

/* ---- one adaptive Golomb-Rice symbol (inverse of EntropyDecodeValue, AlacFile.cs:193-212) ---- */
static void put_symbol(bitwr* w, uint32_t v, int k, uint32_t m, int escape_bits) {
    uint32_t x = v / m, rem = v % m;
    if (x > 8) { /* nine ones, then the raw value */
        put_bits(w, 0x1FF, 9);
        if (escape_bits > 16) {
            put_bits(w, v >> 16, escape_bits - 16);
            put_bits(w, v & 0xFFFF, 16);
        } else {
            put_bits(w, v, escape_bits);
        }
        return;
    }
    for (uint32_t i = 0; i < x; i++) put_bits(w, 1, 1);
    put_bits(w, 0, 1);
    if (k == 1) return;
    if (rem == 0)
        put_bits(w, 0, k - 1); /* decoder reads k bits, sees <= 1 and un-reads one */
    else
        put_bits(w, rem + 1, k);
}

/* ---- Rice stream for one channel (mirror of EntropyRiceDecode, AlacFile.cs:214-252) ---------- */
static int rice_encode(bitwr* w, const int32_t* err, int n, int rss, int init_hist, int kmod, int32_t hist_mult) {
    int32_t history = init_hist;
    int32_t signmod = 0;
    uint32_t kmask = (uint32_t)w_sub(w_shl(1, kmod), 1);
    for (int i = 0; i < n; i++) {
        int kk = 31 - clz_q(w_add(w_sar(history, 9), 3));
        int k = kk < kmod ? kk : kmod; /* :221-222 */
        int32_t r = err[i];
        uint32_t dv = r >= 0 ? (uint32_t)r * 2u : (uint32_t)(-(int64_t)r) * 2u - 1u; /* inverse of :225-226 */
        if (dv < (uint32_t)signmod) return -1; /* cannot happen: a run is always maximal */
        if (k < 1) return -1;
        put_symbol(w, dv - (uint32_t)signmod, k, (1u << k) - 1u, rss);
        signmod = 0;
        history = (int32_t)dv > 0xFFFF ? 0xFFFF
                                       : w_sub(w_add(history, w_mul((int32_t)dv, hist_mult)),
                                               w_sar(w_mul(history, hist_mult), 9)); /* :229 */
        if (history < 128 && i + 1 < n) { /* :231-249 */
            signmod = 1;
            int k2 = clz_q(history) + ((history + 16) / 64) - 24;
            uint32_t z = 0;
            while (i + 1 + (int)z < n && err[i + 1 + z] == 0 && z < 0xFFFF) z++;
            uint32_t m2 = ((1u << (k2 & 31)) - 1u) & kmask;
            if (m2 == 0 || k2 < 1) return -1;
            put_symbol(w, z, k2, m2, 16);
            i += (int)z;
            history = 0;
        }
    }
    return 0;
}

/* ---- forward adaptive predictor (inverse of PredictorDecompressFirAdapt, AlacFile.cs:256-336) -- */
static void predictor_forward(const int32_t* out, int32_t* err, int n, int rss, int32_t* coef, int N, int q) {
    if (n <= 0) return;
    err[0] = out[0];
    if (N == 0) { /* decoder: out = err */
        for (int i = 1; i < n; i++) err[i] = out[i];
        return;
    }
    if (N == 31) { /* :268-282 */
        for (int i = 1; i < n; i++) err[i] = sx(w_sub(out[i], out[i - 1]), rss);
        return;
    }
    for (int i = 1; i <= N && i < n; i++) err[i] = sx(w_sub(out[i], out[i - 1]), rss); /* warm-up :284-293 */
    for (int i = N + 1, b = 0; i < n; i++, b++) { /* :297-334 */
        int32_t sum = 0;
        for (int j = 0; j < N; j++) sum = w_add(sum, w_mul(w_sub(out[b + N - j], out[b]), coef[j]));
        int32_t pred = w_add(w_sar(w_add(w_shl(1, q - 1), sum), q), out[b]);
        int32_t e = sx(w_sub(out[i], pred), rss);
        err[i] = e;
        if (e != 0) { /* identical adaptation, :312-332 */
            int positive = e > 0;
            int p = N - 1;
            while (p >= 0 && (positive ? e > 0 : e < 0)) {
                int32_t val = w_sub(out[b], out[b + N - p]);
                int32_t sg = val < 0 ? -1 : (val > 0 ? 1 : 0);
                int32_t sign = positive ? sg : -sg;
                coef[p] = w_sub(coef[p], sign);
                val = w_mul(val, sign);
                e = w_sub(e, w_mul(w_sar(val, q), N - p));
                p--;
            }
        }
    }
}

/* Levinson-Durbin LPC on one channel; returns quantised coefficients (x[i] ~ sum a_j x[i-1-j]). */
static void lpc_coefs(const int32_t* x, int n, int N, int q, int16_t* out) {
    double r[33], a[33], tmp[33];
    memset(out, 0, sizeof(int16_t) * 32);
    if (N <= 0 || N >= 31 || n <= N + 1) return;
    for (int l = 0; l <= N; l++) {
        double s = 0;
        for (int i = l; i < n; i++) s += (double)x[i] * (double)x[i - l];
        r[l] = s;
    }
    if (r[0] <= 0) return;
    r[0] *= 1.0 + 1e-9;
    double e = r[0];
    memset(a, 0, sizeof(a));
    for (int i = 1; i <= N; i++) {
        double acc = r[i];
        for (int j = 1; j < i; j++) acc -= a[j] * r[i - j];
        double kk = acc / e;
        memcpy(tmp, a, sizeof(a));
        a[i] = kk;
        for (int j = 1; j < i; j++) a[j] = tmp[j] - kk * tmp[i - j];
        e *= (1.0 - kk * kk);
        if (e <= 0) break;
    }
    for (int j = 0; j < N; j++) {
        double c = floor(a[j + 1] * (double)(1 << q) + 0.5);
        if (c > 32767) c = 32767;
        if (c < -32768) c = -32768;
        out[j] = (int16_t)c;
    }
}

size_t alac_synth_max_packet_bytes(uint32_t n, int sample_size, int stereo) {
    size_t ch = stereo ? 2 : 1;
    size_t bits = 23 + 32 + 16 + ch * (16 + 31 * 16) + (size_t)n * ch * 16 + (size_t)n * ch * (9 + 26 + 25) + 3;
    (void)sample_size;
    return bits / 8 + 16;
}

size_t alac_synth_encode_packet(const alac_synth_pkt* d, const int32_t* pcm, uint8_t* out, size_t cap) {
    const int n = (int)d->n;
    const int ch = d->stereo ? 2 : 1;
    const int ss = d->sample_size;
    const int ub = d->escape ? 0 : d->ub;
    if (n <= 0 || n > 16384 || (ss != 16 && ss != 24) || ub > 2 || ss - 8 * ub < 8) return 0;
    bitwr w = {out, cap, 0, 0};
    const int hassize = d->force_hassize || d->n != d->max_samples_per_frame;
    int chfield = d->channels_field >= 0 ? d->channels_field : (d->stereo ? 1 : 0);
    put_bits(&w, (uint32_t)chfield, 3);
    put_bits(&w, 0, 4);
    put_bits(&w, 0, 12);
    put_bits(&w, (uint32_t)hassize, 1);
    put_bits(&w, (uint32_t)ub, 2);
    put_bits(&w, d->escape ? 1u : 0u, 1);
    if (hassize) { put_bits(&w, (uint32_t)n >> 16, 16); put_bits(&w, (uint32_t)n & 0xFFFF, 16); }

    if (d->escape) { /* raw samples, interleaved (AlacFile.cs:500-525 / :665-699) */
        for (int i = 0; i < n; i++)
            for (int c = 0; c < ch; c++) {
                uint32_t v = (uint32_t)pcm[i * ch + c] & (ss == 16 ? 0xFFFFu : 0xFFFFFFu);
                if (ss > 16) { put_bits(&w, v >> (ss - 16), 16); put_bits(&w, v & ((1u << (ss - 16)) - 1u), ss - 16); }
                else put_bits(&w, v, ss);
            }
    } else {
        const int rss = ss - 8 * ub + (d->stereo ? 1 : 0);
        int32_t* A = (int32_t*)malloc(sizeof(int32_t) * (size_t)n * 6);
        if (!A) return 0;
        int32_t *B = A + n, *eA = B + n, *eB = eA + n, *sA = eB + n, *sB = sA + n;
        const uint32_t ubmask = ub ? ((1u << (8 * ub)) - 1u) : 0u;
        for (int i = 0; i < n; i++) {
            int32_t l = pcm[i * ch], r = d->stereo ? pcm[i * ch + 1] : 0;
            sA[i] = (int32_t)((uint32_t)l & ubmask);
            sB[i] = (int32_t)((uint32_t)r & ubmask);
            l >>= 8 * ub; /* arithmetic: decoder rebuilds (l << 8ub) | shift bytes, AlacFile.cs:383-388 */
            r >>= 8 * ub;
            if (d->stereo) {
                if (d->mix_weight != 0) { /* inverse of :349-350 */
                    int32_t diff = w_sub(l, r);
                    A[i] = w_add(r, w_sar(w_mul(diff, d->mix_weight), d->mix_shift));
                    B[i] = diff;
                } else { A[i] = l; B[i] = r; }
            } else { A[i] = l; B[i] = 0; }
        }
        int32_t coef[2][32];
        for (int c = 0; c < ch; c++) {
            int16_t tmp[32];
            memset(tmp, 0, sizeof(tmp));
            int N = d->pred_order[c];
            if (d->coef_mode == 0) lpc_coefs(c == 0 ? A : B, n, N, d->quant[c], tmp);
            else if (d->coef_mode == 1) memcpy(tmp, d->coefs[c], sizeof(tmp));
            for (int j = 0; j < 32; j++) coef[c][j] = tmp[j];
        }
        if (d->stereo) { put_bits(&w, d->mix_shift, 8); put_bits(&w, d->mix_weight, 8); }
        else { put_bits(&w, 0, 8); put_bits(&w, 0, 8); }
        for (int c = 0; c < ch; c++) {
            put_bits(&w, d->pred_type[c] & 15u, 4);
            put_bits(&w, d->quant[c] & 15u, 4);
            put_bits(&w, d->ricemod[c] & 7u, 3);
            put_bits(&w, d->pred_order[c] & 31u, 5);
            for (int j = 0; j < d->pred_order[c]; j++) put_bits(&w, (uint32_t)coef[c][j] & 0xFFFFu, 16);
        }
        if (ub) {
            for (int i = 0; i < n; i++) {
                put_bits(&w, (uint32_t)sA[i], 8 * ub);
                if (d->stereo) put_bits(&w, (uint32_t)sB[i], 8 * ub);
            }
        }
        int bad = 0;
        for (int c = 0; c < ch; c++) {
            predictor_forward(c == 0 ? A : B, c == 0 ? eA : eB, n, rss, coef[c], d->pred_order[c], d->quant[c]);
            int32_t hist_mult = (int32_t)d->ricemod[c] * (d->rice_history_mult / 4); /* :483,:643 */
            if (rice_encode(&w, c == 0 ? eA : eB, n, rss, d->rice_initial_history, d->rice_kmodifier, hist_mult)) bad = 1;
        }
        free(A);
        if (bad) return 0;
    }
    put_bits(&w, 7, 3); /* END element tag, ignored by the reference decoder */
    while (w.bitpos & 7) put_bits(&w, 0, 1);
    if (w.overflow) return 0;
    return w.bitpos >> 3;
}

/* ---- PCM source ------------------------------------------------------------------------------ */
typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t* r) { /* splitmix64 */
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double rng_u(rng_t* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline double rng_gauss(rng_t* r) {
    double u1 = rng_u(r), u2 = rng_u(r);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

void alac_synth_make_pcm(const alac_synth_signal* sig, uint64_t index, int sample_size, int ch, uint32_t n,
                         int32_t* pcm) {
    rng_t r = {sig->seed + index * 0x100000001B3ull};
    rng_next(&r);
    int nsin = 2 + (int)(rng_next(&r) % 2);
    double freq[3], ph[3], amp[3];
    for (int s = 0; s < nsin; s++) {
        freq[s] = 6.283185307179586 * (20.0 + rng_u(&r) * 8000.0) / 44100.0;
        ph[s] = rng_u(&r) * 6.283185307179586;
        amp[s] = pow(2.0, sig->amp_lo_log2 + rng_u(&r) * (sig->amp_hi_log2 - sig->amp_lo_log2));
    }
    uint32_t sil_a = n, sil_b = n;
    if (rng_u(&r) < sig->silence_prob && n > 1) {
        uint32_t len = sig->silence_min + (uint32_t)(rng_next(&r) % (sig->silence_max - sig->silence_min + 1));
        if (len > n) len = n;
        sil_a = (uint32_t)(rng_next(&r) % (n - len + 1));
        sil_b = sil_a + len;
    }
    const double scale = sample_size == 24 ? 256.0 : 1.0;
    const double lim = sample_size == 24 ? 8388607.0 : 32767.0;
    for (uint32_t i = 0; i < n; i++) {
        double l = 0;
        for (int s = 0; s < nsin; s++) l += amp[s] * sin(freq[s] * i + ph[s]);
        double nl = rng_gauss(&r) * sig->noise_sigma, nr = rng_gauss(&r) * sig->noise_sigma;
        double L = (l + nl) * scale, R = (sig->lr_corr * l + nr) * scale;
        if (sample_size == 24) { L += rng_gauss(&r) * 40.0; R += rng_gauss(&r) * 40.0; }
        if (i >= sil_a && i < sil_b) { L = 0; R = 0; }
        L = floor(L + 0.5); R = floor(R + 0.5);
        if (L > lim) L = lim; if (L < -lim - 1) L = -lim - 1;
        if (R > lim) R = lim; if (R < -lim - 1) R = -lim - 1;
        pcm[(size_t)i * ch] = (int32_t)L;
        if (ch == 2) pcm[(size_t)i * ch + 1] = (int32_t)R;
    }
}

/* ---- batch driver ---------------------------------------------------------------------------- */
typedef struct {
    const alac_synth_pkt* descs; const alac_synth_signal* sig; uint64_t first_index;
    uint32_t begin, end;
    uint8_t* tmp; size_t tmp_cap; size_t* tmp_off; /* per packet offset inside tmp */
    uint32_t* sizes; int32_t* pcm_out; uint32_t pcm_slot_ints; int failed;
} sjob_t;

static void* synth_worker(void* arg) {
    sjob_t* j = (sjob_t*)arg;
    size_t used = 0;
    int32_t* pcm = (int32_t*)malloc(sizeof(int32_t) * 16384 * 2);
    if (!pcm) { j->failed = 1; return NULL; }
    for (uint32_t p = j->begin; p < j->end; p++) {
        const alac_synth_pkt* d = &j->descs[p];
        int ch = d->stereo ? 2 : 1;
        if (d->n == 0 || d->n > 16384) { j->failed = 1; break; }
        alac_synth_make_pcm(j->sig, j->first_index + p, d->sample_size, ch, d->n, pcm);
        size_t sz = alac_synth_encode_packet(d, pcm, j->tmp + used, j->tmp_cap - used);
        if (sz == 0) { j->failed = 1; break; }
        j->tmp_off[p] = used;
        j->sizes[p] = (uint32_t)sz;
        used += (sz + 15) & ~(size_t)15;
        if (j->pcm_out) {
            size_t cnt = (size_t)d->n * ch;
            if (cnt > j->pcm_slot_ints) cnt = j->pcm_slot_ints;
            memcpy(j->pcm_out + (size_t)p * j->pcm_slot_ints, pcm, cnt * sizeof(int32_t));
        }
    }
    free(pcm);
    return NULL;
}

size_t alac_synth_make_batch(const alac_synth_pkt* descs, uint32_t n_packets, const alac_synth_signal* sig,
                             uint64_t first_index, int n_threads, uint8_t* blob, size_t blob_cap,
                             uint64_t* offsets, uint32_t* sizes, int32_t* pcm_out, uint32_t pcm_slot_ints) {
    if (n_packets == 0) return 0;
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > n_packets) n_threads = (int)n_packets;
    sjob_t* jobs = (sjob_t*)calloc((size_t)n_threads, sizeof(sjob_t));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    size_t* tmp_off = (size_t*)calloc(n_packets, sizeof(size_t));
    size_t total = 0;
    int failed = 0;
    if (!jobs || !th || !tmp_off) { failed = 1; goto done; }
    for (int t = 0; t < n_threads; t++) {
        sjob_t* j = &jobs[t];
        j->descs = descs; j->sig = sig; j->first_index = first_index;
        j->begin = (uint32_t)((uint64_t)n_packets * t / n_threads);
        j->end = (uint32_t)((uint64_t)n_packets * (t + 1) / n_threads);
        size_t cap = 64;
        for (uint32_t p = j->begin; p < j->end; p++)
            cap += alac_synth_max_packet_bytes(descs[p].n, descs[p].sample_size, descs[p].stereo) + 16;
        j->tmp = (uint8_t*)calloc(cap, 1);
        j->tmp_cap = cap;
        j->tmp_off = tmp_off; j->sizes = sizes; j->pcm_out = pcm_out; j->pcm_slot_ints = pcm_slot_ints;
        if (!j->tmp) failed = 1;
    }
    if (!failed) {
        if (n_threads == 1) synth_worker(&jobs[0]);
        else {
            for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, synth_worker, &jobs[t]);
            for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
        }
        for (int t = 0; t < n_threads && !failed; t++) {
            if (jobs[t].failed) { failed = 1; break; }
            for (uint32_t p = jobs[t].begin; p < jobs[t].end; p++) {
                size_t sz = sizes[p];
                if (total + ((sz + 15) & ~(size_t)15) > blob_cap) { failed = 1; break; }
                memcpy(blob + total, jobs[t].tmp + tmp_off[p], sz);
                offsets[p] = total;
                total += (sz + 15) & ~(size_t)15;
            }
        }
    }
done:
    if (jobs) for (int t = 0; t < n_threads; t++) free(jobs[t].tmp);
    free(jobs); free(th); free(tmp_off);
    return failed ? 0 : total;
}

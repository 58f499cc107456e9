/*
 * alac_oracle.c -- CPU ORACLE (test infrastructure only; see alac_oracle.h for the rules).
 *
 * Scalar C restatement of /root/reference/ALACDecoder/AlacFile.cs (teekay/ALAC.NET).  Every
 * function cites the reference lines it follows.  Quirks of the reference are reproduced on
 * purpose; "parity unpinned" by reference fixtures -- pinned by the
 * hand-derived KATs in tests/test_oracle_kat.py and by the encoder round trip.
 */
#include "alac_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- C# int semantics ----------------------------------------------------------------- */
static inline int32_t cs_shl(int32_t x, int n) { return (int32_t)((uint32_t)x << (n & 31)); }
static inline int32_t cs_sar(int32_t x, int n) { return x >> (n & 31); } /* gcc: arithmetic */
static inline uint32_t cs_shr_u(uint32_t x, int n) { return x >> (n & 31); }
static inline int32_t cs_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t cs_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t cs_mul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }

#define BUFFER_SIZE 16384 /* AlacFile.cs:28 */

/* ---- bit reader: AlacFile.cs:101-152 ---------------------------------------------------- */
typedef struct {
    const uint8_t* buf;
    long len;
    long ibIdx; /* _ibIdx */
    int acc;    /* _inputBufferBitaccumulator */
} bitrd;

/* Reads outside the packet return 0 (the reference would see stale bytes of its reused 80 KiB
 * buffer, AlacContext.cs:64,195: parity is only claimed for packets that do not over-read). */
static inline int32_t rd_byte(const bitrd* b, long i) {
    return (i >= 0 && i < b->len) ? (int32_t)b->buf[i] : 0;
}

/* Readbits16, AlacFile.cs:101-118 */
static int32_t readbits16(bitrd* b, int bits) {
    int32_t part1 = rd_byte(b, b->ibIdx);
    int32_t part2 = rd_byte(b, b->ibIdx + 1);
    int32_t part3 = rd_byte(b, b->ibIdx + 2);
    int32_t result = cs_sar(cs_shl((part1 << 16) | (part2 << 8) | part3, b->acc) & 0x00ffffff, 24 - bits);
    int newAccumulator = b->acc + bits;
    b->ibIdx += newAccumulator >> 3;
    b->acc = newAccumulator & 7;
    return result;
}

/* Readbits, AlacFile.cs:125-129 (high half first; left-to-right evaluation) */
static int32_t readbits(bitrd* b, int bitsParam) {
    int bits = bitsParam <= 16 ? bitsParam : bitsParam - 16;
    int32_t hi = 0;
    if (bitsParam > 16) hi = cs_shl(readbits16(b, 16), bits);
    return hi | readbits16(b, bits);
}

/* Readbit, AlacFile.cs:135-143 */
static int32_t readbit(bitrd* b) {
    int32_t part1 = rd_byte(b, b->ibIdx);
    int32_t result = (cs_shl(part1, b->acc) >> 7) & 1;
    int newAccumulator = b->acc + 1;
    b->ibIdx += newAccumulator / 8;
    b->acc = newAccumulator % 8;
    return result;
}

/* Unreadbits, AlacFile.cs:145-152 (the <0 fix-up is dead code: x & 7 >= 0) */
static void unreadbits(bitrd* b, int bits) {
    int newAccumulator = b->acc - bits;
    b->ibIdx += newAccumulator >> 3;
    b->acc = newAccumulator & 7;
}

static inline long bitpos(const bitrd* b) { return b->ibIdx * 8 + b->acc; }

/* ---- CountLeadingZeros(+Extra): AlacFile.cs:154-191 --------------------------------------
 * Byte-then-nibble search.  Equals clz32 for x != 0 (negative -> 0); returns 40 for x == 0:
 * four loop passes add 32 and the fall-through returns output + 8 (:190). */
static int count_leading_zeros_extra(int32_t curbyteParam, int initialZeroes) {
    int initialCondition = (curbyteParam & 0xf0) == 0;
    int zeroes = initialCondition ? initialZeroes + 4 : initialZeroes;
    int32_t curbyte = !initialCondition ? (curbyteParam >> 4) : curbyteParam;
    if (curbyte & 0x8) return 0 + zeroes;
    if (curbyte & 0x4) return 1 + zeroes;
    if (curbyte & 0x2) return 2 + zeroes;
    if (curbyte & 0x1) return 3 + zeroes;
    return 4 + zeroes;
}

int alac_oracle_count_leading_zeros(int32_t input) {
    int output = 0;
    int32_t curbyte = input >> 24;
    /* pass 0: returnIf x != 0 (whole int, so a negative input returns here), :176 */
    if (curbyte != 0) return count_leading_zeros_extra(curbyte, output);
    output += 8;
    curbyte = input >> 16;
    if ((curbyte & 0xFF) != 0) return count_leading_zeros_extra(curbyte, output);
    output += 8;
    curbyte = input >> 8;
    if ((curbyte & 0xFF) != 0) return count_leading_zeros_extra(curbyte, output);
    output += 8;
    curbyte = input;
    if ((curbyte & 0xFF) != 0) return count_leading_zeros_extra(curbyte, output);
    output += 8;
    return output + 8; /* == 40 */
}

/* ---- EntropyDecodeValue: AlacFile.cs:193-212 ---------------------------------------------- */
static int32_t entropy_decode_value(bitrd* b, int readSampleSize, int k, int32_t riceKmodifierMask) {
    int32_t decodedValue = 0;
    /* incrementUntil: x <= RiceThreshold(8) && Readbit() != 0 ? recurse(x+1) : x   (:196) */
    while (decodedValue <= 8 && readbit(b) != 0) decodedValue++;
    if (decodedValue > 8) {
        return readbits(b, readSampleSize) & (int32_t)cs_shr_u(0xffffffffu, 32 - readSampleSize);
    }
    if (k == 1) return decodedValue;
    int32_t extraBits = readbits(b, k);
    decodedValue = cs_mul(decodedValue, (cs_sub(cs_shl(1, k), 1)) & riceKmodifierMask);
    if (extraBits > 1)
        decodedValue = cs_add(decodedValue, extraBits - 1);
    else
        unreadbits(b, 1);
    return decodedValue;
}

/* ---- EntropyRiceDecode: AlacFile.cs:214-252 -----------------------------------------------
 * Returns 0, or ALAC_ORACLE_OVERRUN when a zero run would index past the reference's
 * 16384-entry scratch (IndexOutOfRangeException at :242).  Runs past outputSize but inside the
 * scratch are tolerated by the reference (they hit stale scratch only) and are clamped here. */
static int entropy_rice_decode(bitrd* b, int32_t* outputBuffer, int outputSize, int readSampleSize,
                               int riceInitialhistory, int riceKmodifier, int32_t riceHistorymult,
                               int32_t riceKmodifierMask) {
    int32_t history = riceInitialhistory;
    int outputCount = 0;
    int32_t signModifier = 0;
    int st = 0;
    int negative_history = 0;
    while (outputCount < outputSize) {
        int initialK = 31 - riceKmodifier - alac_oracle_count_leading_zeros(cs_add(cs_sar(history, 9), 3));
        int k = initialK < 0 ? initialK + riceKmodifier : riceKmodifier;
        int32_t decodedValue = cs_add(entropy_decode_value(b, readSampleSize, k, (int32_t)0xFFFFFFFFu), signModifier);
        int32_t almostFinalValue = cs_add(decodedValue, 1) / 2;
        outputBuffer[outputCount] = (decodedValue & 1) != 0 ? cs_mul(almostFinalValue, -1) : almostFinalValue;
        signModifier = 0;
        history = decodedValue > 0xFFFF
                      ? 0xFFFF
                      : cs_sub(cs_add(history, cs_mul(decodedValue, riceHistorymult)),
                               cs_sar(cs_mul(history, riceHistorymult), 9));
        if (history < 128 && outputCount + 1 < outputSize) {
            /* a negative history (only reachable through 32-bit wrap with exotic multipliers) makes the
             * reference read a negative bit count; outside the supported domain, flagged */
            if (history < 0) negative_history = 1;
            signModifier = 1;
            k = alac_oracle_count_leading_zeros(history) + ((history + 16) / 64) - 24;
            int32_t blockSize = entropy_decode_value(b, 16, k, riceKmodifierMask);
            if (blockSize > 0) {
                for (int32_t j = 0; j < blockSize; j++) {
                    long idx = (long)outputCount + 1 + j;
                    if (idx >= BUFFER_SIZE) { st = ALAC_ORACLE_OVERRUN; break; }
                    if (idx < outputSize) outputBuffer[idx] = 0;
                    else break; /* entries >= outputSize are never read back */
                }
                if ((long)outputCount + blockSize > BUFFER_SIZE - 1) st = ALAC_ORACLE_OVERRUN;
                /* outputCount += blockSize (:244); clamp so the int cannot wrap */
                outputCount = (blockSize > BUFFER_SIZE) ? BUFFER_SIZE : outputCount + blockSize;
            }
            if (blockSize > 0xFFFF) signModifier = 0;
            history = 0;
        }
        outputCount++;
    }
    if (!st && negative_history) st = ALAC_ORACLE_UNSUPPORTED_PARAMS;
    return st;
}

int alac_oracle_rice_decode(const uint8_t* bits, size_t nbytes, int32_t* out, int n, int rss,
                            int init_hist, int kmod, int hist_mult, int* end_bitpos) {
    bitrd b = {bits, (long)nbytes, 0, 0};
    int st = entropy_rice_decode(&b, out, n, rss, init_hist, kmod, hist_mult, cs_sub(cs_shl(1, kmod), 1));
    if (end_bitpos) *end_bitpos = (int)bitpos(&b);
    return st;
}

/* ---- PredictorDecompressFirAdapt: AlacFile.cs:256-336 (in place) --------------------------- */
static inline int32_t sign_extend_rss(int32_t v, int readsamplesize) {
    int bitsmove = 32 - readsamplesize;
    return cs_sar(cs_shl(v, bitsmove), bitsmove);
}

void alac_oracle_predictor(int32_t* buf, int outputSize, int readsamplesize, int32_t* predictorCoefTable,
                           int predictorCoefNum, int predictorQuantitization) {
    int32_t* bufferOut = buf; /* bufferOut = errorBuffer, :260 */
    if (predictorCoefNum == 0) {
        /* :261-267: self Array.Copy = no-op when it does not throw (caller flags n > 4096) */
        return;
    }
    if (predictorCoefNum == 0x1f) { /* :268-282 */
        if (outputSize <= 1) return;
        for (int i = 0; i < outputSize - 1; i++) {
            int32_t prevValue = bufferOut[i];
            int32_t errorValue = buf[i + 1];
            bufferOut[i + 1] = sign_extend_rss(cs_add(prevValue, errorValue), readsamplesize);
        }
        return;
    }
    /* warm-up, :284-293 (reference does not clamp to outputSize; entries >= outputSize are
     * scratch that is never output, so clamping is exact) */
    if (predictorCoefNum > 0) {
        for (int i = 0; i < predictorCoefNum && i + 1 < outputSize; i++) {
            int32_t val = cs_add(bufferOut[i], buf[i + 1]);
            bufferOut[i + 1] = sign_extend_rss(val, readsamplesize);
        }
    }
    if (predictorCoefNum <= 0) return;
    int bufferOutIdx = 0;
    for (int i = predictorCoefNum + 1; i < outputSize; i++) { /* :297-334 */
        int32_t sum = 0;
        int32_t errorVal = buf[i];
        for (int j = 0; j < predictorCoefNum; j++) {
            sum = cs_add(sum, cs_mul(cs_sub(bufferOut[bufferOutIdx + predictorCoefNum - j], bufferOut[bufferOutIdx]),
                                     predictorCoefTable[j]));
        }
        int32_t outval = cs_add(cs_shl(1, predictorQuantitization - 1), sum);
        outval = cs_sar(outval, predictorQuantitization);
        outval = cs_add(cs_add(outval, bufferOut[bufferOutIdx]), errorVal);
        outval = sign_extend_rss(outval, readsamplesize);
        bufferOut[bufferOutIdx + predictorCoefNum + 1] = outval;
        if (errorVal != 0) {
            int positive = errorVal > 0; /* conditionToUse, :316 */
            int predictorNum = predictorCoefNum - 1;
            while (predictorNum >= 0 && (positive ? errorVal > 0 : errorVal < 0)) {
                int32_t val = cs_sub(bufferOut[bufferOutIdx], bufferOut[bufferOutIdx + predictorCoefNum - predictorNum]);
                int32_t sgn = val < 0 ? -1 : (val > 0 ? 1 : 0);
                int32_t sign = positive ? sgn : cs_sub(0, sgn); /* intPosOrNeg, :319,:325 */
                predictorCoefTable[predictorNum] = cs_sub(predictorCoefTable[predictorNum], sign);
                val = cs_mul(val, sign);
                errorVal = cs_sub(errorVal, cs_mul(cs_sar(val, predictorQuantitization), predictorCoefNum - predictorNum));
                predictorNum--;
            }
        }
        bufferOutIdx++;
    }
}

/* ---- Deinterlace16: AlacFile.cs:338-367 ---------------------------------------------------- */
void alac_oracle_deinterlace16(const int32_t* bufferA, const int32_t* bufferB, int32_t* bufferOut, int numchannels,
                               int numsamples, int interlacingShift, int interlacingLeftweight) {
    if (numsamples <= 0) return;
    for (int i = 0; i < numsamples; i++) {
        int32_t left, right;
        if (interlacingLeftweight != 0) {
            int32_t midright = bufferA[i];
            int32_t difference = bufferB[i];
            right = cs_sub(midright, cs_sar(cs_mul(difference, interlacingLeftweight), interlacingShift));
            left = cs_add(right, difference);
        } else {
            left = bufferA[i];
            right = bufferB[i];
        }
        bufferOut[i * numchannels] = left;
        bufferOut[i * numchannels + 1] = right;
    }
}

/* Deinterlace24: AlacFile.cs:369-421, emitted in the canonical int32-per-sample layout
 * (24-bit value sign-extended); the byte-per-int reference layout is the expand function. */
static inline int32_t sx24(int32_t v) { return (int32_t)((uint32_t)v << 8) >> 8; }

static void deinterlace24(const int32_t* bufferA, const int32_t* bufferB, int uncompressedBytes,
                          const int32_t* ubA, const int32_t* ubB, int32_t* bufferOut, int numchannels, int numsamples,
                          int interlacingShift, int interlacingLeftweight) {
    if (numsamples <= 0) return;
    for (int i = 0; i < numsamples; i++) {
        int32_t left, right;
        if (interlacingLeftweight != 0) {
            int32_t midright = bufferA[i];
            int32_t difference = bufferB[i];
            right = cs_sub(midright, cs_sar(cs_mul(difference, interlacingLeftweight), interlacingShift));
            left = cs_add(right, difference);
        } else {
            left = bufferA[i];
            right = bufferB[i];
        }
        if (uncompressedBytes != 0) {
            int32_t mask = (int32_t)~(0xFFFFFFFFu << ((uncompressedBytes * 8) & 31));
            left = cs_shl(left, uncompressedBytes * 8);
            right = cs_shl(right, uncompressedBytes * 8);
            left = left | (ubA[i] & mask);
            right = right | (ubB[i] & mask);
        }
        bufferOut[i * numchannels] = sx24(left);
        bufferOut[i * numchannels + 1] = sx24(right);
    }
}

/* ---- SetInfo: AlacFile.cs:63-93 ------------------------------------------------------------- */
void alac_oracle_set_info(const int32_t* in, int samplesize, int numchannels, alac_oracle_cfg* c) {
    int p = 24; /* six 4-byte skips, :66-71 */
    memset(c, 0, sizeof(*c));
    c->max_samples_per_frame =
        (uint32_t)(cs_add(cs_add(cs_add(cs_shl(in[p], 24), cs_shl(in[p + 1], 16)), cs_shl(in[p + 2], 8)), in[p + 3]));
    p += 4;
    p += 1;                                   /* _setinfo_7A (unused), :74 */
    c->sample_size = (uint8_t)in[p]; p += 1;  /* :76 */
    c->rice_history_mult = (uint8_t)(in[p] & 0xff); p += 1;    /* :78 */
    c->rice_initial_history = (uint8_t)(in[p] & 0xff); p += 1; /* :80 */
    c->rice_kmodifier = (uint8_t)(in[p] & 0xff); p += 1;       /* :82 */
    /* :84-92 hold fields DecodeFrame never reads */
    c->num_channels = (uint8_t)numchannels;
    c->ctor_sample_size = (uint8_t)samplesize;
}

/* ---- DecodeFrame: AlacFile.cs:428-719 ------------------------------------------------------- */
typedef struct {
    int32_t errA[BUFFER_SIZE], errB[BUFFER_SIZE]; /* _predicterrorBufferA/B (outputs alias them, :486,:646,:656) */
    int32_t ubA[BUFFER_SIZE], ubB[BUFFER_SIZE];   /* _uncompressedBytesBufferA/B */
    int32_t coefA[32], coefB[32];                 /* _predictorCoefTable{,A,B}: only N<=31 entries used */
} scratch_t;

static int decode_frame_impl(const alac_oracle_cfg* cfg, scratch_t* s, const uint8_t* packet, size_t packet_size,
                             int32_t* outbuffer, size_t cap, int32_t* out_bytes, int32_t* out_samples) {
    bitrd b = {packet, (long)packet_size, 0, 0};
    const int numchannels = cfg->num_channels;
    const int ctor_ss = cfg->ctor_sample_size ? cfg->ctor_sample_size : cfg->sample_size;
    const int32_t bytespersample = (ctor_ss / 8) * numchannels; /* :19 */
    const int sampleSize = cfg->sample_size;
    const int kmod = cfg->rice_kmodifier;
    const int32_t kmask = cs_sub(cs_shl(1, kmod), 1);
    int32_t outputsamples = (int32_t)cfg->max_samples_per_frame; /* :430 */
    int st = ALAC_ORACLE_OK;

    int32_t channels = readbits(&b, 3);
    int32_t outputsize = cs_mul(outputsamples, bytespersample); /* :436 */
    *out_bytes = outputsize;
    *out_samples = outputsamples;

    if (channels != 0 && channels != 1) return ALAC_ORACLE_UNSUPPORTED_ELEMENT; /* :437,:577,:718 */
    const int stereo = channels == 1;

    readbits(&b, 4);  /* :442,:584 */
    readbits(&b, 12); /* :443,:585 */
    int32_t hassize = readbits(&b, 1);
    int32_t uncompressedBytes = readbits(&b, 2);
    int32_t isnotcompressed = readbits(&b, 1);
    if (hassize != 0) {
        outputsamples = readbits(&b, 32); /* :451,:593 */
        outputsize = cs_mul(outputsamples, bytespersample);
        *out_bytes = outputsize;
        *out_samples = outputsamples;
    }
    /* The switch on sample size comes last in the reference (:527,:701); nothing before it has a
     * visible effect, so it is checked first here. */
    if (sampleSize != 16 && sampleSize != 24) return ALAC_ORACLE_UNSUPPORTED_SAMPLE_SIZE;
    if (numchannels < 1 || numchannels > 2) return ALAC_ORACLE_UNSUPPORTED_ELEMENT;
    if (outputsamples <= 0 || outputsamples > BUFFER_SIZE || (size_t)outputsamples * (size_t)numchannels > cap)
        return ALAC_ORACLE_BAD_SAMPLE_COUNT;
    /* A two-channel element in a one-channel stream: Deinterlace16/24 write out[i] = left, out[i + 1] = right with
     * numchannels == 1 (:353-354, :390-395), so every right sample is overwritten by the next left one and the frame comes
     * out as its LEFT channel (plus one stray int at out[n]).  Reproduced when the slot has room for two channels (the GPU
     * path parks channel A there); otherwise reported as an unsupported element. */
    if (stereo && numchannels < 2 && (size_t)outputsamples * 2 > cap) return ALAC_ORACLE_UNSUPPORTED_ELEMENT;
    if (sampleSize - uncompressedBytes * 8 < 8) return ALAC_ORACLE_UNSUPPORTED_PARAMS;

    int32_t readsamplesize = sampleSize - (uncompressedBytes * 8) + (stereo ? 1 : 0); /* :454,:596 */
    int32_t interlacingShift = 0, interlacingLeftweight = 0;
    int32_t* outA = s->errA;
    int32_t* outB = s->errB;

    if (isnotcompressed == 0) {
        int32_t predictionType[2] = {0, 0}, predictionQuantitization[2] = {0, 0};
        int32_t ricemodifier[2] = {0, 0}, predictorCoefNum[2] = {0, 0};
        int32_t* coefs[2] = {s->coefA, s->coefB};
        if (stereo) {
            interlacingShift = readbits(&b, 8);      /* :599 */
            interlacingLeftweight = readbits(&b, 8); /* :600 (unsigned) */
        } else {
            readbits(&b, 8); /* :459 */
            readbits(&b, 8); /* :460 */
        }
        for (int ch = 0; ch < (stereo ? 2 : 1); ch++) { /* :461-475 / :602-632 */
            predictionType[ch] = readbits(&b, 4);
            predictionQuantitization[ch] = readbits(&b, 4);
            ricemodifier[ch] = readbits(&b, 3);
            predictorCoefNum[ch] = readbits(&b, 5);
            for (int i = 0; i < predictorCoefNum[ch]; i++) {
                int32_t tempPred = readbits(&b, 16);
                if (tempPred > 32767) tempPred = tempPred - 65536;
                coefs[ch][i] = tempPred;
            }
        }
        if (uncompressedBytes != 0) { /* :476-482 / :634-641 */
            for (int i = 0; i < outputsamples; i++) {
                s->ubA[i] = readbits(&b, uncompressedBytes * 8);
                if (stereo) s->ubB[i] = readbits(&b, uncompressedBytes * 8);
            }
        }
        for (int ch = 0; ch < (stereo ? 2 : 1); ch++) { /* :483-496 / :643-661 */
            int32_t* err = ch == 0 ? s->errA : s->errB;
            int32_t histmult = cs_mul(ricemodifier[ch], cfg->rice_history_mult / 4);
            int r = entropy_rice_decode(&b, err, outputsamples, readsamplesize, cfg->rice_initial_history, kmod,
                                        histmult, kmask);
            if (r && !st) st = r;
            if (predictionType[ch] != 0) {
                if (!st) st = ALAC_ORACLE_UNSUPPORTED_PREDTYPE;
                continue;
            }
            if (predictorCoefNum[ch] == 0 && outputsamples > 4096 && !st) st = ALAC_ORACLE_REF_THROWS; /* :264-265 */
            alac_oracle_predictor(err, outputsamples, readsamplesize, coefs[ch], predictorCoefNum[ch],
                                  predictionQuantitization[ch]);
        }
    } else { /* escape: :499-526 / :664-700 */
        for (int i = 0; i < outputsamples; i++) {
            for (int ch = 0; ch < (stereo ? 2 : 1); ch++) {
                int32_t audiobits;
                if (sampleSize <= 16) {
                    audiobits = readbits(&b, sampleSize);
                    int bitsmove = 32 - sampleSize;
                    audiobits = cs_sar(cs_shl(audiobits, bitsmove), bitsmove);
                } else {
                    int32_t m = 1 << (24 - 1);
                    audiobits = readbits(&b, 16);
                    audiobits = cs_shl(audiobits, sampleSize - 16);
                    audiobits = audiobits | readbits(&b, sampleSize - 16);
                    int32_t x = audiobits & ((1 << 24) - 1);
                    audiobits = (x ^ m) - m;
                }
                (ch == 0 ? outA : outB)[i] = audiobits;
            }
        }
        uncompressedBytes = 0;
        interlacingShift = 0;
        interlacingLeftweight = 0;
    }

    if (bitpos(&b) > (long)packet_size * 8 && !st) st = ALAC_ORACLE_OVERRUN;

    if (!stereo) { /* :527-575 */
        for (int i = 0; i < outputsamples; i++) {
            int32_t sample = outA[i];
            if (sampleSize == 24) {
                if (uncompressedBytes != 0) {
                    sample = cs_shl(sample, uncompressedBytes * 8);
                    int32_t mask = (int32_t)~(0xFFFFFFFFu << ((uncompressedBytes * 8) & 31));
                    sample = sample | (s->ubA[i] & mask);
                }
                sample = sx24(sample);
            }
            outbuffer[i * numchannels] = sample;
            /* :540 / :563-565: the "next channel" is zeroed; with numchannels==1 that is the next
             * frame's slot, overwritten by the next iteration (the stray last one is slot slack). */
            if (numchannels == 2) outbuffer[i * numchannels + 1] = 0;
        }
    } else if (sampleSize == 16) { /* :705 */
        alac_oracle_deinterlace16(outA, outB, outbuffer, numchannels, outputsamples, interlacingShift,
                                  interlacingLeftweight);
    } else { /* :710 */
        deinterlace24(outA, outB, uncompressedBytes, s->ubA, s->ubB, outbuffer, numchannels, outputsamples,
                      interlacingShift, interlacingLeftweight);
    }
    return st;
}

int alac_oracle_decode_frame(const alac_oracle_cfg* cfg, const uint8_t* packet, size_t packet_size, int32_t* pcm,
                             size_t pcm_capacity, int32_t* out_bytes, int32_t* out_samples) {
    scratch_t* s = (scratch_t*)malloc(sizeof(scratch_t));
    int32_t ob = 0, os = 0;
    if (!s) return -1;
    memset(s, 0, sizeof(*s));
    int st = decode_frame_impl(cfg, s, packet, packet_size, pcm, pcm_capacity, &ob, &os);
    if (out_bytes) *out_bytes = ob;
    if (out_samples) *out_samples = os;
    free(s);
    return st;
}

size_t alac_oracle_expand_reference_layout(const alac_oracle_cfg* cfg, const int32_t* pcm, int32_t n_samples,
                                           int32_t* ref) {
    size_t total = (size_t)n_samples * cfg->num_channels;
    if (cfg->sample_size != 24) {
        memcpy(ref, pcm, total * sizeof(int32_t));
        return total;
    }
    for (size_t i = 0; i < total; i++) { /* :390-395, :555-557 */
        ref[3 * i + 0] = pcm[i] & 0xFF;
        ref[3 * i + 1] = (pcm[i] >> 8) & 0xFF;
        ref[3 * i + 2] = (pcm[i] >> 16) & 0xFF;
    }
    return 3 * total;
}

/* FormatSamples: AlacContext.cs:214-256 */
size_t alac_oracle_format_samples(int bps, const int32_t* src, int32_t samcnt, uint8_t* dst) {
    size_t counter = 0, counter2 = 0;
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

/* ---- batch driver ------------------------------------------------------------------------- */
typedef struct {
    const alac_oracle_cfg* cfgs; uint32_t n_cfgs;
    const uint8_t* blob; const uint64_t* offsets; const uint32_t* sizes; const uint16_t* cfg_idx;
    uint32_t begin, end;
    int32_t* pcm_out; uint32_t slot_ints;
    int32_t* out_bytes; int32_t* out_samples; int32_t* status;
} job_t;

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    scratch_t* s = (scratch_t*)calloc(1, sizeof(scratch_t));
    if (!s) return NULL;
    for (uint32_t p = j->begin; p < j->end; p++) {
        uint32_t ci = j->cfg_idx ? j->cfg_idx[p] : 0;
        int32_t ob = 0, os = 0, st;
        if (ci >= j->n_cfgs) {
            st = ALAC_ORACLE_UNSUPPORTED_PARAMS;
        } else {
            st = decode_frame_impl(&j->cfgs[ci], s, j->blob + j->offsets[p], j->sizes[p],
                                   j->pcm_out + (size_t)p * j->slot_ints, j->slot_ints, &ob, &os);
        }
        if (j->out_bytes) j->out_bytes[p] = ob;
        if (j->out_samples) j->out_samples[p] = os;
        if (j->status) j->status[p] = st;
    }
    free(s);
    return NULL;
}

int alac_oracle_decode_batch(const alac_oracle_cfg* cfgs, uint32_t n_cfgs, const uint8_t* blob,
                             const uint64_t* offsets, const uint32_t* sizes, const uint16_t* cfg_idx,
                             uint32_t n_packets, int32_t* pcm_out, uint32_t slot_ints, int32_t* out_bytes,
                             int32_t* out_samples, int32_t* status, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > n_packets) n_threads = n_packets ? (int)n_packets : 1;
    job_t* jobs = (job_t*)calloc((size_t)n_threads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    if (!jobs || !th) { free(jobs); free(th); return -1; }
    for (int t = 0; t < n_threads; t++) {
        job_t j = {cfgs, n_cfgs, blob, offsets, sizes, cfg_idx,
                   (uint32_t)((uint64_t)n_packets * t / n_threads), (uint32_t)((uint64_t)n_packets * (t + 1) / n_threads),
                   pcm_out, slot_ints, out_bytes, out_samples, status};
        jobs[t] = j;
    }
    if (n_threads == 1) {
        worker(&jobs[0]);
    } else {
        for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, worker, &jobs[t]);
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    }
    free(jobs);
    free(th);
    return 0;
}

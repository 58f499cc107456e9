// alac_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the ALAC frame-decode path.
//
// What it computes: AlacFile.DecodeFrame (reference ALACDecoder/AlacFile.cs:428-719) for a batch of
// packets: element header parse, adaptive Golomb-Rice entropy decode (EntropyRiceDecode :214-252 =
// basterdised_rice_decompress), adaptive sign-LMS FIR reconstruction (PredictorDecompressFirAdapt
// :256-336 = predictor_decompress_fir_adapt) and mid/side un-mixing + store (Deinterlace16/24
// :338-421), integer only, bit-exact with the reference's C# int semantics.
//
// One kernel family lives here, the "two-pass" kernels: 8 (or 16) packets per workgroup, channel A decoded
// for real in pass 0 (its end is where B starts: no Rice-only pre-scan), parked in the packet's own output slot, and
// un-mixed with B in pass 1; one-channel and uncompressed packets finish in pass 0.  Per workgroup: an entropy wave
// (bitstream staged in per-stream LDS rings, branch-free speculative Rice units -> LDS queue of code values), an output
// wave (turns the code values into what the FIR step wants, un-mixes, merges shift bytes, stores the PCM coalesced, refills
// the rings) and one FIR wave per 8 streams (8 lanes per stream, 1 / 2 / 4 taps per lane, DPP reductions and scans).
// No MFMA: the path is integer/branchy, not a contraction.  No floating point anywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "alac_kernels.h"
#include "alac_device.h"
#include "alac_diag.h"

using namespace alacdev;

namespace {

// Output store.  out_format 0: the canonical int32 per sample.  out_format 1: what AlacContext.Read hands
// out -- FormatSamples (AlacContext.cs:214-256) fused into the store: 16-bit streams the low 16 bits little-endian
// (:231-242), 24-bit streams the three bytes of the sample (:244-252 over the byte-per-int layout).
__device__ __forceinline__ void store_sample(const alac_decode_params& p, const Meta& m, int32_t* pcm_slot, int64_t idx,
                                             int val) {
    if (p.out_format == 0) {
        pcm_slot[idx] = val;
    } else if (m.ss == 16) {
        reinterpret_cast<uint16_t*>(pcm_slot)[idx] = (uint16_t)val;
    } else {
        uint8_t* b = reinterpret_cast<uint8_t*>(pcm_slot) + idx * 3;
        b[0] = (uint8_t)val;
        b[1] = (uint8_t)(val >> 8);
        b[2] = (uint8_t)(val >> 16);
    }
}

#ifndef ALAC_ENTROPY_PRIO
#define ALAC_ENTROPY_PRIO 3
#endif

template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        T u = (T)__shfl_xor((int)v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}

// One speculative unit (SPEC_UNIT samples) of the entropy wave.  Tiers, cheapest first:
//   1  rice_spec_step       plain values only
//   2  rice_spec_step_z     + zero runs in progress / signModifier pending   (digital silence)
//   1E rice_spec_step_esc   plain values and escape codes, raw value <= 23 bits (loud / noisy 16-bit content)
//      rice_spec_step_esc_wide  the same for wider raw values (24-bit streams; the window slides by up to two dwords)
//   3  rice_spec_step_full  escape codes of any width together with zero runs / signModifier
// A unit whose lanes met something its tier cannot do is restored from its snapshot and retried higher; a NEW run
// symbol (history < 128 after a value) is beyond all tiers: the function then returns false with the state
// restored, and the caller decodes the unit with rice_step.  After an escape was seen the wave stays on the escape
// tier for ESC_HOLD / FULL_HOLD clean units (a failed cheaper attempt costs a whole unit; measured on cfg2:
// hold 0 / 1 / 2 / 4 / 8 -> 0.828 / 0.815 / 0.825 / 0.843 / 0.858 ms).
#ifndef ALAC_FULL_HOLD
#define ALAC_FULL_HOLD 1
#endif
constexpr int FULL_HOLD = ALAC_FULL_HOLD;

#ifndef ALAC_ESC_HOLD
#define ALAC_ESC_HOLD 1
#endif
constexpr int ESC_HOLD = ALAC_ESC_HOLD;
// Escape codes that keep coming (a loud passage whose peaks escape every few units: 90 of a packet's 512 units in the slowest
// workgroups of cfg2) are cheaper on the escape tier throughout than as one failed plain unit each: when a plain unit fails
// within ESC_NEAR units of the last escape code, the wave stays on the escape tier until ESC_LONG units in a row were clean.
// (Measured, profiles/experiments/r3_adaptive_escape_hold.txt: 8 / 8 in the small-batch build -2 % on cfg2 at 4096 packets;
// the other builds keep NEAR = 0, i.e. the fixed hold.)
#ifndef ALAC_ESC_NEAR
#define ALAC_ESC_NEAR 0
#endif
#ifndef ALAC_ESC_LONG
#define ALAC_ESC_LONG 1
#endif
// What keeps the hold of the narrow escape tier alive: a unary prefix above this in the unit (8: an escape code, nothing less;
// the small-batch build: 7 -- a prefix of eight ones says that the next escape code is close: -0.5 .. -1 %, r3_chunk_units.txt 8)
#ifndef ALAC_ESC_PREFIX
#define ALAC_ESC_PREFIX 8
#endif
// Streams beyond the narrow plain step's arithmetic (rice_spec_step: full-scale noise) stay on the wide plain step for WIDE_HOLD
// units at a time.
#ifndef ALAC_WIDE_HOLD
#define ALAC_WIDE_HOLD 32
#endif
// ALAC_NARROW=0 leaves the narrow step out (the wide one is then THE plain step).  The small-batch build does: at <= 4096
// packets a launch is as long as its slowest workgroup, which sits on the escape tier most of the time, and the extra tier's
// code costs that workgroup more than its few plain units gain (cfg2 at 4096 packets: 0.693 -> 0.698 ms with it).
#ifndef ALAC_NARROW
#define ALAC_NARROW 1
#endif
constexpr bool NARROW_PLAIN = ALAC_NARROW != 0;
struct TierState {
    int wide;        // units still to decode with the wide plain step
    int full_left;   // units still to decode on an escape-capable tier
    int hold;        // clean units the escape tier waits for before handing back to the plain tier
    int since;       // units since the last escape code
};

// After a tier-1-like pass: the parked lanes (see spec_unit) drop out of the verdict and get their state back.
template <bool WANT_R, int QSTRIDE, int CNT = SPEC_UNIT>
__device__ __forceinline__ void spec_unpark(Rice& rs, const Rice& snap, bool parked, uint32_t& xmax, int& hmin, int* q) {
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(parked) != 0, 0)) {
        xmax = parked ? 0u : xmax;
        hmin = parked ? 0x7FFFFFFF : hmin;
        if (parked) {
            rs.w0 = snap.w0; rs.w1 = snap.w1; rs.w2 = snap.w2;
            rs.cur = snap.cur; rs.ra = snap.ra; rs.hist = snap.hist;
            rs.zrun = snap.zrun - CNT;
            if (WANT_R) {
#pragma unroll
                for (int ii = 0; ii < CNT; ii++) q[ii * QSTRIDE] = 0;
            }
        }
    }
}

// Queue stores of a unit: with a row per stream (QSTRIDE 1) four values go out in ONE 16-byte store -- every LDS instruction
// is an issue slot of the wave that has none to spare.
template <int QSTRIDE>
__device__ __forceinline__ void spec_store(int* q, int ii, int (&acc)[4], int r) {
    if (QSTRIDE == 1) {
        acc[ii & 3] = r;
        if ((ii & 3) == 3) *reinterpret_cast<int4*>(q + (ii & ~3)) = make_int4(acc[0], acc[1], acc[2], acc[3]);
    } else {
        q[ii * QSTRIDE] = r;
    }
}

template <bool WANT_R, int QSTRIDE, bool RAW = false>
__device__ __forceinline__ bool spec_unit(Rice& rs, TierState& ts, const RiceCfg& c, uint32_t ring, int* q, SpecStats& st) {
    const Rice snap = rs;
    // A stream whose zero run covers the whole unit is PARKED: its lanes run the plain code with everybody else (on
    // whatever their window shows -- harmless, see rice_spec_step), are left out of the unit's verdict and get their state
    // back afterwards, minus SPEC_UNIT zeros: one silent stream does not keep the other seven on tier 2.
    const bool parked = rs.zrun >= SPEC_UNIT;
    const bool special = __builtin_amdgcn_ballot_w64(rs.nforce == 0u && !parked) != 0;
    uint32_t xmax = 0;
    int hmin = 0x7FFFFFFF;
    int acc[4] = {0, 0, 0, 0};
    if (ts.full_left == 0) {
        bool newrun = false, sawesc = false;
        bool wide = !NARROW_PLAIN || ts.wide > 0;
        if (NARROW_PLAIN && __builtin_expect(!special && !wide, 1)) {
            // the narrow plain step (rice_spec_step): the hot path of the whole kernel
            uint32_t cmin = 32;
            rs.hist += RICE_PLAIN_BIAS;            // (the narrow step's form of the history)
#pragma unroll
            for (int ii = 0; ii < SPEC_UNIT; ii++) {
                const int r = rice_spec_step<WANT_R, RAW>(rs, c, ring, xmax, hmin, cmin);
                if (WANT_R) spec_store<QSTRIDE>(q, ii, acc, r);
            }
            rs.hist -= RICE_PLAIN_BIAS;
            hmin -= RICE_PLAIN_BIAS;
            spec_unpark<WANT_R, QSTRIDE>(rs, snap, parked, xmax, hmin, q);
            newrun = __builtin_amdgcn_ballot_w64(hmin < 128) != 0;
            // an escape code -- or a history, a k or a value beyond the narrow step's arithmetic
            sawesc = __builtin_amdgcn_ballot_w64(xmax > 8u || (!parked && !rice_narrow_ok(c, xmax, cmin))) != 0;
            if (__builtin_expect(!newrun && !sawesc, 1)) { SPEC_COUNT(plain_ok); ts.since++; return true; }
            // no escape code, then: the same unit again with the wide plain step (what the narrow one made of such a stream's
            // history means nothing: no verdict on runs from it)
            if (sawesc && __builtin_amdgcn_ballot_w64(xmax > 8u) == 0) {
                rs = snap;
                xmax = 0;
                hmin = 0x7FFFFFFF;
                ts.wide = ALAC_WIDE_HOLD;
                wide = true;
                SPEC_COUNT(fail_range);
            }
        }
        if (!special && wide) {
            uint32_t vmax = 0;
#pragma unroll
            for (int ii = 0; ii < SPEC_UNIT; ii++) {
                const int r = rice_spec_step_wide<WANT_R, RAW>(rs, c, ring, xmax, hmin, vmax);
                if (WANT_R) spec_store<QSTRIDE>(q, ii, acc, r);
            }
            spec_unpark<WANT_R, QSTRIDE>(rs, snap, parked, xmax, hmin, q);
            newrun = __builtin_amdgcn_ballot_w64(hmin < 128) != 0;
            // an escape code -- or a value whose history update needs the clamp the plain steps leave out
            sawesc = __builtin_amdgcn_ballot_w64(xmax > 8u || (!parked && vmax > 0xFFFFu)) != 0;
            // back to the narrow step once every stream is well inside its range again (looked at every WIDE_HOLD units)
            if (NARROW_PLAIN && --ts.wide == 0 &&
                __builtin_amdgcn_ballot_w64(!parked && __builtin_clz((uint32_t)rs.hist + RICE_PLAIN_BIAS) <= c.cfloor) != 0)
                ts.wide = ALAC_WIDE_HOLD;
            if (NARROW_PLAIN) SPEC_COUNT(wide_units);
            if (__builtin_expect(!newrun && !sawesc, 1)) { if (!NARROW_PLAIN) SPEC_COUNT(plain_ok); ts.since++; return true; }
        } else if (special) {
#pragma unroll
            for (int ii = 0; ii < SPEC_UNIT; ii++) {
                const int r = rice_spec_step_z<WANT_R, RAW>(rs, c, ring, xmax, hmin);
                if (WANT_R) spec_store<QSTRIDE>(q, ii, acc, r);
            }
            rs.nforce = (rs.zrun > 0 || rs.signmod != 0) ? 0u : 0xFFFFFFFFu;
            newrun = __builtin_amdgcn_ballot_w64(hmin < 128) != 0;
            sawesc = __builtin_amdgcn_ballot_w64(xmax > 8u) != 0;
            if (__builtin_expect(!newrun && !sawesc, 1)) { SPEC_COUNT(z_units); ts.since++; return true; }
        }
        rs = snap;
        if (newrun) { SPEC_COUNT(fail_run); return false; }          // no tier can do it
        SPEC_COUNT(fail_esc);
        ts.full_left = 1;                  // escapes: go on with an escape-capable tier
        ts.hold = ts.since < ALAC_ESC_NEAR ? ALAC_ESC_LONG : ESC_HOLD;
        xmax = 0;
        hmin = 0x7FFFFFFF;
    }
    if (!special && __builtin_amdgcn_ballot_w64(c.rss > 23) == 0) {   // tier 1E (rss is per stream: ask the whole wave)
#pragma unroll
        for (int ii = 0; ii < SPEC_UNIT; ii++) {
            const int r = rice_spec_step_esc<WANT_R, RAW>(rs, c, ring, xmax, hmin);
            if (WANT_R) spec_store<QSTRIDE>(q, ii, acc, r);
        }
        spec_unpark<WANT_R, QSTRIDE>(rs, snap, parked, xmax, hmin, q);
        const bool sawesc = __builtin_amdgcn_ballot_w64(xmax > (unsigned)ALAC_ESC_PREFIX) != 0;
        ts.full_left = sawesc ? ts.hold : ts.full_left - 1;
        ts.since = sawesc ? 0 : ts.since + 1;
        SPEC_COUNT(esc_units);
    } else if (!special) {                 // tier 1E for wide raw values (24-bit streams)
        uint32_t w3 = lds_load(((rs.ra + 4u) & RING_MASK) | ring);
#pragma unroll
        for (int ii = 0; ii < SPEC_UNIT; ii++) {
            const int r = rice_spec_step_esc_wide<WANT_R, RAW>(rs, w3, c, ring, xmax, hmin);
            if (WANT_R) spec_store<QSTRIDE>(q, ii, acc, r);
        }
        spec_unpark<WANT_R, QSTRIDE>(rs, snap, parked, xmax, hmin, q);
        const bool sawesc = __builtin_amdgcn_ballot_w64(xmax > 8u) != 0;
        ts.full_left = sawesc ? ts.hold : ts.full_left - 1;
        ts.since = sawesc ? 0 : ts.since + 1;
        SPEC_COUNT(esc_units);
    } else {                               // tier 3
        uint32_t w3 = lds_load(((rs.ra + 4u) & RING_MASK) | ring);
#pragma unroll
        for (int ii = 0; ii < SPEC_UNIT; ii++) {
            const int r = rice_spec_step_full<WANT_R, RAW>(rs, w3, c, ring, xmax, hmin);
            if (WANT_R) spec_store<QSTRIDE>(q, ii, acc, r);
        }
        rs.nforce = (rs.zrun > 0 || rs.signmod != 0) ? 0u : 0xFFFFFFFFu;
        const bool sawesc = __builtin_amdgcn_ballot_w64(xmax > 8u) != 0;
        ts.full_left = sawesc ? FULL_HOLD : ts.full_left - 1;
        ts.since = sawesc ? 0 : ts.since + 1;
        SPEC_COUNT(full_units);
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(hmin < 128) != 0, 0)) {
        rs = snap;
        SPEC_COUNT(late_run);
        return false;
    }
    return true;
}

// A whole chunk as ONE unit of the escape tier, for a wave that is on that tier to stay (loud passages: the slowest workgroups
// of a small batch spend most of their units there; 24-bit streams without shift bytes are coded in escape codes throughout) --
// the code between two units costs this wave a quarter of the steps themselves.  Taken when the hold has at least
// ALAC_ESC_CHUNK units to go (0: never) and no stream is in a zero run that ends inside the chunk; a new run symbol (not seen
// in a loud passage) restores the state and returns false: the chunk is then decoded unit by unit.  Raw values of up to 23
// bits only (the same for wider ones -- the second launch's cfg3 -- measured level at the BASELINE sizes).
// (Measured, profiles/experiments/r3_chunk_units.txt: cfg2 at 4096 packets 0.697 -> 0.676 ms; the slowest workgroup 1.66 M ->
// 1.57 M cycles, an escape-tier unit 599 -> 280 cycles above a plain one.  The same for the PLAIN tier lost 1.5 .. 6 %.)
#ifndef ALAC_ESC_CHUNK
#define ALAC_ESC_CHUNK 0
#endif
template <int QSTRIDE, int CNT>
__device__ __forceinline__ bool esc_chunk(Rice& rs, TierState& ts, const RiceCfg& c, uint32_t ring, int* q, SpecStats& st) {
    const bool parked = rs.zrun >= CNT;
    if (__builtin_amdgcn_ballot_w64(rs.nforce == 0u && !parked) != 0) return false;
    if (__builtin_amdgcn_ballot_w64(c.rss > 23) != 0) return false;
    const Rice snap = rs;
    uint32_t xmax = 0;
    int hmin = 0x7FFFFFFF;
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int ii = 0; ii < CNT; ii++) {
        const int r = rice_spec_step_esc<true, true>(rs, c, ring, xmax, hmin);
        spec_store<QSTRIDE>(q, ii, acc, r);
    }
    spec_unpark<true, QSTRIDE, CNT>(rs, snap, parked, xmax, hmin, q);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(hmin < 128) != 0, 0)) {
        rs = snap;
        return false;
    }
    const bool sawesc = __builtin_amdgcn_ballot_w64(xmax > (unsigned)ALAC_ESC_PREFIX) != 0;
    ts.full_left = sawesc ? ts.hold : max(ts.full_left - CNT / SPEC_UNIT, 0);
    ts.since = sawesc ? 0 : ts.since + CNT / SPEC_UNIT;
    DIAG_ONLY(st.esc_units += CNT / SPEC_UNIT;)
    return true;
}

// value of `v` in lane `src` (a lane mirrors itself when src == its own id)
__device__ __forceinline__ int mirror_i(int v, int src) { return __shfl(v, src, 64); }

// ===================================================================================================================
// The "two-pass" kernels: 8 packets per workgroup (16 in the dense arrangement), three working waves per 8 streams
// (entropy, output, FIR; which wave takes which role is decided from the SIMDs they landed on, see ab_kernel_body).
// Pass 0 decodes channel A of all packets for real (entropy wave -> queue of code values -> output wave: pre-digested
// residuals -> FIR wave -> output wave), and the output wave parks the reconstructed samples in the upper half of the
// packet's own output slot; where pass 0 ends IS where channel B starts, so there is no Rice-only pre-scan.  Pass 1 decodes
// channel B the same way and the output wave un-mixes it with the parked A samples (read back one chunk ahead) and stores
// the PCM.  One-channel and uncompressed packets finish in pass 0.
// The first launch (alac_decode_ab*_kernel) takes the groups of 8 packets whose streams have LPC order 1..8 (one tap per
// lane of the FIR wave; the dense arrangement: 1..16) and flags the others (alac_decode_params::ab_flags) for
// alac_decode_ab32_kernel, launched behind it: the same code with two taps per lane (orders 9..16) or four (any order, the
// delta mode 31, order 0).
// Parking place: ints [n, 2n) of the slot (slot_ints >= 2n for two channels).  The final stores of sample i touch
// at most int 2i+1 (int32 output) or byte 6i+5 (packed), always below the parked samples not yet consumed.
//
// Barriers: every wave executes nchunks + 2 per pass.  Between barriers b and b+1 the entropy wave decodes chunk b+1, the
// output wave converts chunk b (resq[b & 1] -> xq[b & 1]) and stores chunk b-2 (outq[b & 1]), the FIR wave reconstructs
// chunk b-1 (xq[(b-1) & 1] -> outq[(b-1) & 1]): every queue is written in one interval and read in the next.
// ===================================================================================================================

// The rings are topped up by the OUTPUT wave (which idles most of the time) instead of the entropy wave (which is the
// critical one): the entropy wave publishes how far each stream has read, at every chunk barrier; the output wave then
// loads the next 128 bytes of every stream that has room, swaps them and writes them into the ring before the next
// barrier.  What it wrote after barrier c is read after barrier c+1 at the earliest; the ring is kept at least 880 bytes
// ahead of the reader, who consumes at most 118 per chunk.
#ifndef ALAC_AB_CHUNK
#define ALAC_AB_CHUNK 32
#endif
// Layout of the queue of code values (entropy wave -> output wave).  1: a row per stream, the 16 values of a speculative unit
// side by side, stored four at a time (12 LDS instructions fewer per unit for the wave that has no issue slot to spare): the
// latency-bound small-batch build, -3 % (cfg2 / cfg4 at 4096 packets).  0: a row per sample, one store per value: what the
// other builds measured best with (the conversion's reads are conflict-free; cfg4 at 8192 packets +2 % with rows per stream,
// cfg2 at 32768 +1 %).  profiles/experiments/r3_queue_rows.txt
#ifndef ALAC_QROWS
#define ALAC_QROWS 0
#endif
constexpr bool QROWS = ALAC_QROWS != 0;
#ifndef ALAC_L16_MAX_GROUPS
#define ALAC_L16_MAX_GROUPS 256     // groups of 8 packets in the launch (= one workgroup per CU) up to which orders above 16 take two FIR waves
#endif
constexpr int AB_CHUNK = ALAC_AB_CHUNK;   // samples per barrier (16 were measured too: the critical wave pays its per-chunk overhead twice as often)
// NS = streams (packets) per workgroup: 8 (one entropy wave of 8 lanes per stream, one FIR wave, one output wave) or 16 (the
// "dense" arrangement for big batches: ONE entropy wave serves 16 streams with 4 lanes each -- its instructions are shared
// by twice as many packets -- next to two FIR waves of 8 streams and one output wave for all 16).
template <int NS> struct XqSel { typedef XQ type; };
template <> struct XqSel<16> { typedef XQ8 type; };
template <int NS>
struct AbSharedT {
    typedef typename XqSel<NS>::type Xq;
    uint32_t rings[NS][RING_BYTES / 4];
    // entropy wave -> output wave: the unsigned Rice code values dv.  QROWS: [stream][sample], rows 16-byte aligned and, with
    // 4 ints of padding, spread over the banks; else [sample][stream]
    int resq[2][QROWS ? NS : AB_CHUNK][QROWS ? AB_CHUNK + 4 : NS];
    Xq xq[2][AB_CHUNK][NS + 1];          // output wave -> FIR waves; column NS is never written and stays zero: what a
                                         // switched-off or finished stream is fed
    int outq[2][AB_CHUNK / 8][NS * 8 + 64];   // FIR wave w -> output wave (lanes 64 w ..), 8 outputs per stream per 8 samples; the 64
                                              // ints behind them: where FIR lanes that hold no tap below 8 write (no branch around a store)
    int dummy[QROWS ? AB_CHUNK + 4 + 64 + 4 : AB_CHUNK * NS + 64];   // where the entropy wave's lanes without a stream of their own write
    uint32_t ring_next[NS];    // entropy wave -> output wave: Rice::next of the stream at the last barrier
    uint32_t ring_filled[NS];  // entropy wave -> output wave at the start of a pass: bytes staged by rice_init
    uint32_t ring_on[NS];      // stream switched on in this pass
};

// The inverse of r = (dv >> 1) ^ -(dv & 1) (AlacFile.cs:225-226) for what rice_step hands back: the queue carries dv.
__device__ __forceinline__ int ab_zigzag(int r) { return (int)(((uint32_t)r << 1) ^ (uint32_t)(r >> 31)); }

// One entropy pass over stream `g` of every lane group.  Returns the bit position after the last symbol; *flags_out
// collects rice_step's flags.
template <int NS>
__device__ __forceinline__ uint32_t ab_entropy_pass(const alac_decode_params& p, AbSharedT<NS>& sh, const Meta& m, const RiceCfg& rc, int init_hist,
                                    uint32_t startbit, bool stream_on, int g, int sub, int lane, int nchunks, int* flags_out
                                    DIAG_ONLY(, SpecStats& st)) {
    constexpr int S = NS, LPS = 64 / NS;
    const int n_row = stream_on ? m.n : 0;
    int flags = 0;
    // every stream's first value is coded against the initial history (a small k): as a rule an escape code -- the pass
    // starts on the escape tier instead of failing its first plain unit
    TierState ts;
    ts.wide = 0;
    ts.full_left = 1;
    ts.hold = ESC_HOLD;
    ts.since = ALAC_ESC_NEAR;
#ifndef ALAC_DIAG
    SpecStats st;
#endif
    Rice rs;
    rs.w0 = rs.w1 = rs.w2 = 0; rs.next = 12; rs.hist = 0; rs.signmod = 0; rs.zrun = 0; rs.nforce = 0;
    rs.ra = rs.ra_sync = lds_addr(sh.rings[g]);
    rs.cur = rice_cursor(0, rs.ra);
    uint32_t filled = 0;
    // Lanes with nothing to decode SHADOW the longest stream of the wave (same ring, same state, results thrown away), so
    // that the wave stays in lock step on the speculative units: streams that are switched off from the start, and --
    // from the chunk after its last sample on -- every stream that is shorter than the longest (ragged batches: the last
    // packet of a file, mixed frame lengths).  Only the chunk in which a stream ends is decoded by the generic loop.
    const int nmax = __builtin_amdgcn_readfirstlane(wave_max(n_row));
    const uint64_t longest = __builtin_amdgcn_ballot_w64(stream_on && n_row == nmax);
    const int srcmax = longest ? (int)__builtin_ctzll(longest) : lane;
    bool real = stream_on;                       // this lane decodes its own stream
    const int src = real ? lane : srcmax;
    RiceCfg mc;
    mc.kmod = mirror_i(rc.kmod, src);
    mc.kmask = (1u << (mc.kmod & 31)) - 1u;      // (1 << kb) - 1 with C#'s five-bit shift count (AlacFile.cs:483)
    mc.hist_mult = mirror_i(rc.hist_mult, src);
    mc.rss = mirror_i(rc.rss, src);
    rice_cfg_finish(mc);
    int n_eff = mirror_i(m.n, src);
    const int ih = mirror_i(init_hist, src);
    const uint32_t sb = (uint32_t)mirror_i((int)startbit, src);
    const uint32_t* mringp = sh.rings[mirror_i(g, src)];
    uint32_t mring = lds_addr(mringp);
    int nmin = longest ? __builtin_amdgcn_readfirstlane(-wave_max(-n_eff)) : 0;
    if (nmax > 0) {
        rice_init<LPS>(rs, filled, sb, ih, sh.rings[g], m.base, m.limit, sub, stream_on);
        if (!stream_on) {
            const uint32_t d0 = rs.next - 12u;
            rs.w0 = mringp[(d0 & RING_MASK) >> 2];
            rs.w1 = mringp[((d0 + 4u) & RING_MASK) >> 2];
            rs.w2 = mringp[((d0 + 8u) & RING_MASK) >> 2];
            rs.cur = rice_cursor((int)(rs.cur & 31u), mring | ((d0 + 8u) & RING_MASK));
            rs.ra = rs.ra_sync = mring | ((d0 + 8u) & RING_MASK);
        }
    }
    if (sub == 0) {   // hand the ring over to the output wave (it reads this after the first barrier of the pass)
        sh.ring_on[g] = stream_on ? 1u : 0u;
        sh.ring_filled[g] = filled;
        sh.ring_next[g] = rs.next;
    }
    uint32_t endpos = 0;
    int endflags = 0;
    bool ended = false;
    int c = 0;
    while (c < nchunks) {
        // ---- between stretches: streams that have ended become shadows of the longest one (rare) ----
        {
            const int i0 = c * AB_CHUNK;
            const bool fin = real && i0 >= n_row;
            if (__builtin_amdgcn_ballot_w64(fin) != 0) {
                if (fin) {
                    endpos = rice_bitpos(rs);         // (rice_sync ran at the end of the last chunk)
                    endflags = flags;
                    ended = true;
                    real = false;
                    if (sub == 0) sh.ring_on[g] = 0u; // no more refills for this ring
                }
                const int s2 = fin ? srcmax : lane;
                rs.w0 = (uint32_t)mirror_i((int)rs.w0, s2);
                rs.w1 = (uint32_t)mirror_i((int)rs.w1, s2);
                rs.w2 = (uint32_t)mirror_i((int)rs.w2, s2);
                rs.cur = (uint32_t)mirror_i((int)rs.cur, s2);
                rs.ra = (uint32_t)mirror_i((int)rs.ra, s2);
                rs.ra_sync = (uint32_t)mirror_i((int)rs.ra_sync, s2);
                rs.next = (uint32_t)mirror_i((int)rs.next, s2);
                rs.hist = mirror_i(rs.hist, s2);
                rs.signmod = mirror_i(rs.signmod, s2);
                rs.zrun = mirror_i(rs.zrun, s2);
                rs.nforce = (uint32_t)mirror_i((int)rs.nforce, s2);
                mc.kmod = mirror_i(mc.kmod, s2);
                mc.kmask = (1u << (mc.kmod & 31)) - 1u;
                mc.hist_mult = mirror_i(mc.hist_mult, s2);
                mc.rss = mirror_i(mc.rss, s2);
                mring = (uint32_t)mirror_i((int)mring, s2);
                n_eff = mirror_i(n_eff, s2);
                nmin = __builtin_amdgcn_readfirstlane(-wave_max(-n_eff));
            }
        }
        // ---- a stretch of chunks up to the one in which the next stream ends: nothing below changes who shadows whom ----
        const int my_stop = real ? (n_row + AB_CHUNK - 1) / AB_CHUNK : 0x7FFFFFFF;
        const int c_stop = min(max(__builtin_amdgcn_readfirstlane(-wave_max(-my_stop)), c + 1), nchunks);
        const RiceCfg kc = mc;
        const uint32_t kring = mring & ~(uint32_t)RING_MASK;   // (a no-op: rings are 1 KiB aligned -- but the compiler must see it to fold the address into one v_bitop3)
        const int kn = n_eff, knmin = nmin;
        const bool kreal = real;
        // The chunks that lie wholly inside every stream (all but the last one or two of a stretch) in a loop of their own: nothing
        // to decide per chunk (this wave pays for every block boundary: see the FIR waves).
        {
            int* const qa = (sub == 0 && kreal) ? (QROWS ? &sh.resq[0][g][0] : &sh.resq[0][0][g]) : &sh.dummy[QROWS ? 4 * (lane & 15) : lane];
            const int qodd = (sub == 0 && kreal) ? AB_CHUNK * S + (QROWS ? 4 * S : 0) : 0;
            constexpr int QS1 = QROWS ? 1 : S;        // distance of two samples of a stream in the queue
            const int cf_end = min(c_stop, knmin / AB_CHUNK);
            for (; c < cf_end; c++) {
                const int i0 = c * AB_CHUNK;
                int* q = qa + (c & 1) * qodd;
                bool whole = false;     // the whole chunk as one unit of the escape tier (esc_chunk)
                if (ALAC_ESC_CHUNK && ts.full_left >= ALAC_ESC_CHUNK)
                    whole = esc_chunk<QS1, AB_CHUNK>(rs, ts, kc, kring, q, st);
                if (!whole) {
#pragma unroll
                    for (int u = 0; u < AB_CHUNK; u += SPEC_UNIT) {
                        const bool redo = !spec_unit<true, QS1, true>(rs, ts, kc, kring, q + u * QS1, st);
                        if (__builtin_expect(redo, 0)) {
                            SPEC_COUNT(redo);
                            for (int ii = 0; ii < SPEC_UNIT; ii++) {
                                const int r = rice_step(rs, kc, kn - 1 - (i0 + u + ii), i0 + u + ii, &flags, kring);
                                q[(u + ii) * QS1] = ab_zigzag(r);
                            }
                        }
                    }
                }
                rice_sync(rs);
                if (sub == 0) sh.ring_next[g] = rs.next;
                wg_sync();  // chunk c is ready for the output wave
            }
        }
        for (; c < c_stop; c++) {
            const int i0 = c * AB_CHUNK;
            constexpr int QS1 = QROWS ? 1 : S;
            int* q = (sub == 0 && kreal) ? (QROWS ? &sh.resq[c & 1][g][0] : &sh.resq[c & 1][0][g]) : &sh.dummy[QROWS ? 4 * (lane & 15) : lane];
            if (i0 < nmax) {
                // (a stream may END with the chunk: the reference reads no run symbol behind a stream's last sample, whatever the
                // history -- AlacFile.cs:231 -- while the speculative step reports one; such a unit is then decoded by rice_step,
                // which knows.  The generic loop below costs 570 cycles per sample.)
                const bool fast_chunk = i0 + AB_CHUNK <= knmin;
                if (fast_chunk) {
#pragma unroll
                    for (int u = 0; u < AB_CHUNK; u += SPEC_UNIT) {
                        const bool redo = !spec_unit<true, QS1, true>(rs, ts, kc, kring, q + u * QS1, st);
                        if (redo) {
                            SPEC_COUNT(redo);
                            for (int ii = 0; ii < SPEC_UNIT; ii++) {
                                const int r = rice_step(rs, kc, kn - 1 - (i0 + u + ii), i0 + u + ii, &flags, kring);
                                q[(u + ii) * QS1] = ab_zigzag(r);
                            }
                        }
                    }
                } else {      // some stream ends in this chunk (or one sample after it): shadows step with their source
                    const int qstride = (sub == 0 && kreal) ? QS1 : 0;
                    for (int ii = 0; ii < AB_CHUNK; ii++) {
                        const int i = i0 + ii;
                        int r = 0;
                        if (i < kn) r = rice_step(rs, kc, kn - 1 - i, i, &flags, kring);
                        q[ii * qstride] = ab_zigzag(r);
                    }
                }
                rice_sync(rs);
                if (sub == 0) sh.ring_next[g] = rs.next;
            }
            wg_sync();  // chunk c is ready for the output wave
        }
    }
    wg_sync();      // the two barriers behind the last chunk (every wave executes nchunks + 2 per pass): the chunk is
    wg_sync();      // converted, then reconstructed
    rice_sync(rs);
    *flags_out = stream_on ? (ended ? endflags : flags) : 0;
    return ended ? endpos : rice_bitpos(rs);
}

template <int NS>
__device__ __forceinline__ void ab_entropy_wave(const alac_decode_params& p, uint32_t pkt0, int lane, AbSharedT<NS>& sh, int nch0, int nch1) {
    constexpr int LPS = 64 / NS;
    const int g = lane / LPS, sub = lane % LPS;
    const uint32_t pkt = pkt0 + (uint32_t)g;
    const bool valid = pkt < p.n_packets;
    alacgpu_cfg_dev cfg;
    const Meta m = parse_meta(p, pkt, 0, valid, cfg);
    const Meta mb = parse_meta(p, pkt, 1, valid, cfg);
    if (valid && sub == 0) {
        if (p.out_bytes) p.out_bytes[pkt] = m.out_bytes;
        if (p.out_samples) p.out_samples[pkt] = m.n;
    }
    const bool compressed = valid && m.status == 0 && !m.esc;
    RiceCfg rc;
    rc.kmod = cfg.rice_kmodifier;
    rc.kmask = (1u << (cfg.rice_kmodifier & 31)) - 1u;
    rc.hist_mult = m.ricemod * (cfg.rice_history_mult / 4);
    rc.rss = m.rss;
    int flags_a = 0, flags_b = 0;
    uint32_t end_a = m.ricebit, end_b = m.ricebit;
    DIAG_ONLY(SpecStats st;)
    // one copy of the pass's code for both channels (two would not fit the instruction cache next to the other waves' code)
    for (int ph = 0; ph < (nch1 > 0 ? 2 : 1); ph++) {
        int fl = 0;
        rc.hist_mult = (ph ? mb.ricemod : m.ricemod) * (cfg.rice_history_mult / 4);
        const uint32_t end = ab_entropy_pass<NS>(p, sh, m, rc, cfg.rice_initial_history, ph ? end_a : m.ricebit,
                                             compressed && (ph == 0 || m.stereo), g, sub, lane, ph ? nch1 : nch0, &fl DIAG_ONLY(, st));
        if (ph == 0) { end_a = end_b = end; flags_a = fl; }
        else { end_b = end; flags_b = fl; }
    }
    DIAG_ONLY(if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + 8 * blockIdx.x;
        d[1] = ((unsigned long long)st.plain_ok << 32) | (unsigned)st.fail_esc;
        d[4] = ((unsigned long long)st.z_units << 32) | (unsigned)st.esc_units;
        d[5] = ((unsigned long long)st.fail_run << 48) | ((unsigned long long)st.redo << 32);
        d[6] = ((unsigned long long)st.full_units << 48) | ((unsigned long long)st.late_run << 32);
        d[7] = ((unsigned long long)st.wide_units << 32) | (unsigned)st.fail_range;
        d[2] = clock64();
    })
    // ---- status, in the reference's control-flow order (same as the oracle) ----
    if (valid && sub == 0) {
        int st = m.status;
        if (st == 0 && !m.esc) {
            const int nch = m.stereo ? 2 : 1;
            for (int c = 0; c < nch && st == 0; c++) {
                const int fl = c == 0 ? flags_a : flags_b;
                const int pt = c == 0 ? m.predtype : mb.predtype;
                const int Nc = c == 0 ? m.N : mb.N;
                if (fl & 1) st = ALACGPU_ST_OVERRUN_D;
                else if (fl & 2) st = ALACGPU_ST_UNSUPPORTED_PARAMS_D;
                else if (pt != 0) st = ALACGPU_ST_UNSUPPORTED_PREDTYPE_D;
                else if (Nc == 0 && m.n > 4096) st = ALACGPU_ST_REF_THROWS_D;
            }
            const uint32_t last = m.stereo ? end_b : end_a;
            if (st == 0 && last > m.size_bits_end) st = ALACGPU_ST_OVERRUN_D;
        } else if (st == 0 && m.esc) {
            const uint32_t last = m.rawbit + (uint32_t)(m.n * (m.stereo ? 2 : 1) * m.ss);
            if (last > m.size_bits_end) st = ALACGPU_ST_OVERRUN_D;
        }
        p.status[pkt] = st;
    }
}

// The 32 steady-state steps of a chunk as one straight-line block (a lone wave pays for every block boundary); I = step
// number inside the chunk, I % T = the register rotation phase (0 again at every multiple of 8).
template <int T, bool WIDE, bool SPECIAL, int NS, int L, int I>
__device__ __forceinline__ void ab_fir_block(FirB<T>& f, const typename XqSel<NS>::type* q, typename XqSel<NS>::type x, int* oq,
                                             const int (&oidx)[T]) {
    if constexpr (I < AB_CHUNK) {
        // the next step's residual is fetched one step ahead (an LDS read takes a good hundred cycles to come back)
        const typename XqSel<NS>::type xn = q[(I + 1 < AB_CHUNK ? I + 1 : I) * (NS + 1)];
        firb_step<T, WIDE, SPECIAL, I % T, L>(f, x);
        if constexpr ((I & 7) == 7) {
#pragma unroll
            for (int r = 0; r < T; r++) oq[(I >> 3) * (NS * 8 + 64) + oidx[r]] = f.h[r];
        }
        ab_fir_block<T, WIDE, SPECIAL, NS, L, I + 1>(f, q, xn, oq, oidx);
    }
}

// FIR wave (alac_device.h: the blocked layout), T taps per lane.  L = 8 lanes per stream: the two parities of a row of 16 lanes
// hold the SAME channel of two different packets, so one wave serves the 8 streams of a pass; w: which block of 8 streams of
// the workgroup this wave serves (always 0 when NS == 8).  L = 16 (orders above 16): a row per stream, TWO waves share the 8
// streams of the workgroup (w = 0 / 1: streams 0..3 / 4..7): half the taps per lane, so a step of either wave is a little more
// than half as long -- and the FIR step's serial chain is what such a workgroup waits for.
template <int T, bool WIDE, bool SPECIAL, int NS, int L = 8>
__device__ __forceinline__ void ab_fir_wave(const alac_decode_params& p, uint32_t pkt0, int w, int lane, AbSharedT<NS>& sh, int ph, int nchunks) {
    constexpr int QS = NS + 1;
    const int row = lane >> 4, l = lane & 15, par = L == 16 ? 0 : l & 1, jl = L == 16 ? l : l >> 1;
    const int g = L == 16 ? 4 * w + row : 8 * w + 2 * row + par;
    const uint32_t pkt = pkt0 + (uint32_t)g;
    const bool valid = pkt < p.n_packets;
    alacgpu_cfg_dev cfg;
    const Meta m = parse_meta(p, pkt, ph, valid, cfg);
    const bool stream_on = valid && m.status == 0 && !m.esc && (ph == 0 || m.stereo);
    const int n_row = stream_on ? m.n : 0;
    // A one-channel element with a prediction type other than 0: the reference skips the predictor without a word and hands
    // out _outputsamplesBufferA (AlacFile.cs:484-496) -- which, once any compressed frame has been decoded, IS the residual
    // buffer (:486: the predictor returns its input array): the un-predicted residuals come out.  Reproduced as order 0;
    // the status (3) stays as a warning.  (A two-channel element throws, :650/:660: status 3, output unspecified.)
    const bool unpredicted = stream_on && !m.stereo && m.predtype != 0;
    const int N = unpredicted ? 0 : m.N;
    const bool gen = stream_on && N >= 1 && N <= 30;         // the general mode (:297-334)
    FirB<T> f;
    f.q = stream_on ? m.q : 1;
    f.rss = stream_on ? m.rss : 16;
    f.qmask = (1 << f.q) - 1;
    f.bias = 1 << (f.rss - 1);
    f.N = stream_on ? N : 0;
    f.n0 = stream_on && N == 0;
    f.delta = stream_on && N == 31;
    f.rnd0 = (gen && jl == 0) ? m.rnd : 0;
    f.base = f.prev = f.bias;
    int oidx[T];           // where this lane's registers go in the output queue: tap t < 8 -> the lane 2 t + par of the row
#pragma unroll
    for (int r = 0; r < T; r++) {
        const int t = T * jl + r;
        const bool tap = gen && t < N;
        f.h[r] = f.bias;
        f.c[r] = tap ? (int)(int16_t)peek_bits(m.base, m.limit, m.coefbit + 16u * t, 16) : 0;   // :466-475
        f.tlo[r] = tap ? -1 : 0;
        f.thi[r] = tap ? 1 : 0;
        f.w[r] = tap ? (uint32_t)(N - t) : 0u;
        // tap t < 8 -> the lane 2 t + par of the row (what the output wave's lane wants); the others: the spare ints behind
        oidx[r] = t < 8 ? 16 * (g >> 1) + 2 * t + (g & 1) : NS * 8 + lane;
    }
    const int tl = gen ? N - 1 : 0;
    f.bpaddr = ((lane & 48) + (L == 16 ? tl / T : 2 * (tl / T) + par)) * 4;
    f.bsel = tl % T;
    const int nmax = __builtin_amdgcn_readfirstlane(wave_max(n_row));
    const int nmin_on = __builtin_amdgcn_readfirstlane(-wave_max(stream_on ? -n_row : (int)0x80000001));
    const int c_fast = nmax > 0 ? min(nchunks, nmin_on / AB_CHUNK) : 0;      // chunks that lie wholly inside every switched-on stream
    constexpr int QODD = AB_CHUNK * QS;
    wg_sync();      // barrier 0 of the pass: chunk 0 decoded, not yet converted
    for (int c = 0; c < nchunks; c++) {
        wg_sync();  // chunk c is converted
        const int i0 = c * AB_CHUNK;
        int* const oq = &sh.outq[c & 1][0][0];
        // a stream that is switched off or has ended runs along on zeros (its outputs are not stored)
        const typename XqSel<NS>::type* const q = &sh.xq[0][0][(i0 < n_row) ? g : NS] + (c & 1) * QODD;
        // The common chunk -- behind the warm-up chunk and wholly inside every stream that is still running -- is 32 steady-state
        // steps in one block.  While it lies inside every switched-on stream there is nothing to ask the wave (c < c_fast);
        // behind the end of the shortest one (ragged batches) it must not be the chunk in which a stream ends.
        bool fast = c >= 1;
        if (__builtin_expect(c >= c_fast, 0))
            fast = fast && i0 + AB_CHUNK <= nmax && __builtin_amdgcn_ballot_w64(i0 < n_row && i0 + AB_CHUNK > n_row) == 0;
        if (__builtin_expect(fast, 1)) {
            ab_fir_block<T, WIDE, SPECIAL, NS, L, 0>(f, q, q[0], oq, oidx);
            continue;
        }
#pragma unroll 1
        for (int half = 0; half < AB_CHUNK / 8; half++) {
            const int ih = i0 + 8 * half;
            if (ih < nmax) {
#pragma unroll 1
                for (int ii = 0; ii < 8; ii++) {
                    const int i = ih + ii;
                    firb_step_masked<T, L>(f, q[(8 * half + ii) * QS].err, i, i < n_row);
                }
            }
#pragma unroll
            for (int r = 0; r < T; r++) oq[half * (NS * 8 + 64) + oidx[r]] = f.h[r];
        }
    }
    wg_sync();  // final barrier of the pass
}

// 24-bit streams: merge the sample's shift bytes (AlacFile.cs:390-395 / :476-482, :634-641) and sign-extend to 24 bits
// (:555-557).  The shift bytes of sample i -- one field of 8 ub bits per channel, the channels side by side -- sit at bit
// ubit + i * channels * 8 ub of the packet: at most 32 bits at any bit alignment, i.e. inside two big-endian dwords.  The
// output wave fetches those two dwords ONE STEP AHEAD of the sample (ub_fetch; like the parked channel A), so that the
// loads' latency -- a global load is a microsecond or two -- lies behind a chunk barrier instead of in front of every store
// (cfg3: the output wave's exposed loads were a tenth of the 8-packet arrangement's time and a quarter of the dense one's).
struct UbWin { uint32_t hi, lo; };
__device__ __forceinline__ bool ab_has_ub(const Meta& m) { return m.ss == 24 && m.ub != 0 && !m.esc; }
__device__ __forceinline__ UbWin ub_fetch(const Meta& m, int i, bool want) {
    UbWin w;
    w.hi = w.lo = 0;
    if (want) {
        const uint32_t bp = m.ubit + (uint32_t)(i * (m.stereo ? 2 : 1) * 8 * m.ub);
        const int64_t d = (int64_t)(bp >> 5) * 4;
        w.hi = load_be32(m.base, d, m.limit);
        w.lo = load_be32(m.base, d + 4, m.limit);
    }
    return w;
}
// AHEAD false: the window is fetched here and now (the dense arrangement, see AbOutBlock::ub_step)
template <bool AHEAD>
__device__ __forceinline__ int ab_finish24(const Meta& m, UbWin w, int val, int i, int chan) {
    if (m.ss != 24) return val;
    if (m.ub != 0 && !m.esc) {
        if (!AHEAD) w = ub_fetch(m, i, true);
        const uint32_t bp = m.ubit + (uint32_t)(i * (m.stereo ? 2 : 1) * 8 * m.ub);
        const uint32_t off = (bp & 31u) + (uint32_t)(chan * 8 * m.ub);                    // <= 31 + 16
        const uint64_t both = (((uint64_t)w.hi << 32) | w.lo) << off;
        const uint32_t field = (uint32_t)(both >> 32) >> (32 - 8 * m.ub);
        val = (int)(((uint32_t)val << (8 * m.ub)) | field);
    }
    return __builtin_amdgcn_sbfe(val, 0, 24);
}

// Ring refill service of the output wave (see AbShared): lane group r = lane >> 3 serves stream r, 16 bytes per lane.
template <int NS>
struct AbRefill {
    static constexpr int ROUNDS = AB_CHUNK / 16;   // up to 8 bytes per sample per stream per chunk (a sample consumes at most 59 bits)
    // a sample costs at most 59 bits (a run-length symbol, 9 + 16, and an escaped value, 9 + 25): never fall behind
    static_assert(AB_CHUNK * 59 <= ROUNDS * 128 * 8, "the ring refill must keep up with the worst-case consumption");
    // what the entropy wave reads during a chunk was staged before the barrier that started it: the refill issued one barrier
    // earlier left the ring >= RING_BYTES - 12 - 127 bytes ahead of the reader's position then, one chunk's consumption ago
    static_assert(RING_BYTES - 12 - 127 - (AB_CHUNK * 59 + 7) / 8 >= (AB_CHUNK * 59 + 7) / 8 + 16,
                  "a chunk this long can outrun a ring this small");
    const uint8_t* base;
    int64_t limit;
    AbSharedT<NS>& sh;
    uint32_t* ring;
    int r, sub;
    uint32_t filled = 0, cnt = 0;
    uint4 v[ROUNDS];
    __device__ AbRefill(const alac_decode_params& p, uint32_t pkt0, int w, int lane, AbSharedT<NS>& sh_) : sh(sh_) {
        r = 8 * w + (lane >> 3);
        sub = lane & 7;
        ring = sh.rings[r];
        const uint32_t pk = pkt0 + (uint32_t)r;
        base = p.blob;
        limit = 0;
        if (pk < p.n_packets) {                // as parse_meta: 16-byte aligned-down packet start, bytes up to the packet's end
            const uint64_t off = p.offsets[pk];
            const uint64_t al = off - (off & 15u);
            base = p.blob + al;
            limit = min((int64_t)p.blob_limit - (int64_t)al, (int64_t)(off & 15u) + (int64_t)p.sizes[pk]);
        }
    }
    // after a barrier: look at how far the stream has read and load what fits (first: a pass starts, take over `filled`)
    __device__ void issue(bool first) {
        if (first) filled = sh.ring_filled[r];
        const uint32_t next = sh.ring_next[r];
        const bool on = sh.ring_on[r] != 0;
        cnt = 0;
#pragma unroll
        for (int k = 0; k < ROUNDS; k++) {
            v[k] = make_uint4(0, 0, 0, 0);
            if (on && filled + (uint32_t)(k + 1) * 128u <= (next - 12u) + RING_BYTES) {
                v[k] = load16_clamped(base, (int64_t)filled + (int64_t)k * 128 + sub * 16, limit);
                cnt = (uint32_t)(k + 1);
            }
        }
    }
    // before the next barrier: byte-swap into the ring
    __device__ void commit() {
#pragma unroll
        for (int k = 0; k < ROUNDS; k++) {
            if ((uint32_t)k < cnt) {
                const uint32_t off = filled + (uint32_t)k * 128u + (uint32_t)sub * 16u;
                uint4 w = v[k];
                asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(w.z), "+v"(w.w));   // keep the swap (and the wait for the load) here
                *reinterpret_cast<uint4*>(&ring[(off & RING_MASK) >> 2]) =
                    make_uint4(__builtin_bswap32(w.x), __builtin_bswap32(w.y), __builtin_bswap32(w.z), __builtin_bswap32(w.w));
            }
        }
        filled += cnt * 128u;
    }
};

// The output work for one block of 8 streams (block w of the workgroup): what happens between two chunk barriers of pass 0
// (pass0_step) and of pass 1 (pass1_step).  One output wave serves one block (8-packet workgroups) or both blocks of a
// 16-packet workgroup, one after the other, between the same two barriers.
// Two lane -> stream mappings live here: the FIR wave's (lane 2 j + par of a row holds out[last - j] of stream 2 row + par),
// for the queue of reconstructed samples, and a linear one (stream lane & 7, sample lane >> 3; or the other way round, QROWS) for the conversion of the code
// values and for the ring refill.
template <int NS>
struct AbOutBlock {
    const alac_decode_params& p;
    AbSharedT<NS>& sh;
    int w, lane, j, g;
    Meta m;
    int n_out, bias;
    bool two_pass;
    int32_t* pcm_slot;
    int32_t* park;
    AbRefill<NS> rf;
    int a_next[AB_CHUNK / 8];
    UbWin ub_next[AB_CHUNK / 8];   // the shift-byte windows of the samples stored at the NEXT step
    int cs, cq[2];          // conversion: this lane's stream and the quantiser masks (1 << q) - 1 of its two channels
    __device__ AbOutBlock(const alac_decode_params& p_, uint32_t pkt0, int w_, int lane_, AbSharedT<NS>& sh_)
        : p(p_), sh(sh_), w(w_), lane(lane_), rf(p_, pkt0, w_, lane_, sh_) {
        const int row = lane >> 4, l = lane & 15, par = l & 1;
        j = l >> 1;
        g = 8 * w + 2 * row + par;
        const uint32_t pkt = pkt0 + (uint32_t)g;
        const bool valid = pkt < p.n_packets;
        alacgpu_cfg_dev cfg;
        m = parse_meta(p, pkt, 0, valid, cfg);
        n_out = (valid && m.status == 0) ? m.n : 0;
        bias = 1 << (m.rss - 1);
        two_pass = n_out > 0 && m.stereo && !m.esc;   // A is parked in pass 0 and finished in pass 1
        pcm_slot = p.pcm_out + (int64_t)pkt * p.slot_ints;
        park = p.park ? p.park + (int64_t)pkt * p.park_stride : pcm_slot + m.n;
#pragma unroll
        for (int h = 0; h < AB_CHUNK / 8; h++) { a_next[h] = 0; ub_next[h].hi = ub_next[h].lo = 0; }
        cs = 8 * w + (QROWS ? lane >> 3 : lane & 7);
        const uint32_t cpkt = pkt0 + (uint32_t)cs;
        const bool cvalid = cpkt < p.n_packets;
        const Meta ca = parse_meta(p, cpkt, 0, cvalid, cfg);
        const Meta cb = parse_meta(p, cpkt, 1, cvalid, cfg);
        cq[0] = (1 << ca.q) - 1;
        cq[1] = (1 << cb.q) - 1;
    }
    // Shift-byte windows for the samples of chunk b-1, fetched one step AHEAD of their stores (they wait in 8 registers across
    // the barrier).  8-packet arrangement only: the dense arrangement's kernel sits at its 128-register budget -- with these
    // 16 more (its output wave serves two blocks) the compiler spilled, and not only in the output wave: cfg2 at 32768 packets
    // 2.56 -> 3.24 ms -- so there the window is fetched where it is used, as in round 2.
    static constexpr bool UB_AHEAD = NS == 8;
    __device__ __forceinline__ void ub_step(int b, int nch, bool wanted, UbWin (&ub_cur)[AB_CHUNK / 8]) {
        if (!UB_AHEAD) return;
#pragma unroll
        for (int h = 0; h < AB_CHUNK / 8; h++) ub_cur[h] = ub_next[h];
        if (!wanted) return;
#pragma unroll
        for (int half = 0; half < AB_CHUNK / 8; half++) {
            const int ih = (b - 1) * AB_CHUNK + 8 * half;
            const int cnt = min(8, n_out - ih);
            ub_next[half] = ub_fetch(m, ih + cnt - 1 - j, b >= 1 && b <= nch && j < cnt);
        }
    }
    // after barrier b of a pass: chunk b's code values -> what the FIR step wants (four residuals per lane)
    __device__ __forceinline__ void convert(int b, int ph) {
        const auto& src = sh.resq[b & 1];
        typename XqSel<NS>::type (*dst)[NS + 1] = sh.xq[b & 1];
#pragma unroll
        for (int k = 0; k < AB_CHUNK / 8; k++) {
            const int i = QROWS ? (lane & 7) + 8 * k : (lane >> 3) + 8 * k;
            typename XqSel<NS>::type x;
            xq_from_code(x, (uint32_t)(QROWS ? src[cs][i] : src[i][cs]), cq[ph]);
            dst[i][cs] = x;
        }
    }
    // after barrier b of pass 0: chunk b-2's outputs are in the queue
    __device__ __forceinline__ void pass0_step(int b, int nch0) {
        if (b < nch0) rf.issue(b == 0);
        UbWin ub_cur[AB_CHUNK / 8];
#pragma unroll
        for (int h = 0; h < AB_CHUNK / 8; h++) ub_cur[h].hi = ub_cur[h].lo = 0;
        ub_step(b, nch0, !two_pass && ab_has_ub(m), ub_cur);
        if (b < nch0) convert(b, 0);
        if (b >= 2) {
            const int c = b - 2;
#pragma unroll
            for (int half = 0; half < AB_CHUNK / 8; half++) {
                const int ih = c * AB_CHUNK + 8 * half;                    // start of the FIR layout's block
                const int cnt = min(8, n_out - ih);
                if (j >= cnt) continue;
                const int mine = sh.outq[c & 1][half][64 * w + lane] - bias;   // lane (2 t + par) holds out[last - t] of its stream
                if (m.esc) {                                            // uncompressed: raw samples, both channels now
                    const int i = ih + j;
                    const int nch = m.stereo ? 2 : 1;
                    for (int ch = 0; ch < nch; ch++) {
                        const uint32_t bp = m.rawbit + (uint32_t)((i * nch + ch) * m.ss);
                        int val = __builtin_amdgcn_sbfe((int)peek_bits(m.base, m.limit, bp, m.ss), 0, m.ss);
                        if (m.ss == 24) val = __builtin_amdgcn_sbfe(val, 0, 24);
                        if (ch < m.nc) store_sample(p, m, pcm_slot, (int64_t)i * m.nc + ch, val);
                    }
                    if (!m.stereo && m.nc > 1) store_sample(p, m, pcm_slot, (int64_t)i * m.nc + 1, 0);
                } else {
                    const int i = ih + cnt - 1 - j;
                    if (two_pass) {
                        park[i] = mine;
                    } else {                                            // one channel: done
                        store_sample(p, m, pcm_slot, (int64_t)i * m.nc, ab_finish24<UB_AHEAD>(m, ub_cur[half], mine, i, 0));
                        if (m.nc > 1) store_sample(p, m, pcm_slot, (int64_t)i * m.nc + 1, 0);
                    }
                }
            }
        }
        if (b < nch0) rf.commit();
    }
    // after barrier b of pass 1: B's chunk b-2 arrives, A comes back from its parking place (loaded one step ahead)
    __device__ __forceinline__ void pass1_step(int b, int nch1) {
        if (b < nch1) rf.issue(b == 0);
        UbWin ub_cur[AB_CHUNK / 8];
#pragma unroll
        for (int h = 0; h < AB_CHUNK / 8; h++) ub_cur[h].hi = ub_cur[h].lo = 0;
        ub_step(b, nch1, two_pass && ab_has_ub(m), ub_cur);
        if (b < nch1) convert(b, 1);
        int a_cur[AB_CHUNK / 8];
#pragma unroll
        for (int h = 0; h < AB_CHUNK / 8; h++) a_cur[h] = a_next[h];
#pragma unroll
        for (int half = 0; half < AB_CHUNK / 8; half++) {                      // A for chunk b-1 (used after the next barrier)
            const int ih = (b - 1) * AB_CHUNK + 8 * half;
            const int cnt = min(8, n_out - ih);
            a_next[half] = (two_pass && b >= 1 && b <= nch1 && j < cnt) ? park[ih + cnt - 1 - j] : 0;
        }
        if (b >= 2) {
            const int c = b - 2;
#pragma unroll
            for (int half = 0; half < AB_CHUNK / 8; half++) {
                const int ih = c * AB_CHUNK + 8 * half;
                const int cnt = min(8, n_out - ih);
                if (!two_pass || j >= cnt) continue;
                const int i = ih + cnt - 1 - j;
                const int a = a_cur[half], bb = sh.outq[c & 1][half][64 * w + lane] - bias;
                int left, right;
                if (m.mixweight != 0) {                                 // AlacFile.cs:346-357 / :377-388
                    right = wsub(a, wmul(bb, m.mixweight) >> (m.mixshift & 31));
                    left = wadd(right, bb);
                } else {
                    left = a;
                    right = bb;
                }
                store_sample(p, m, pcm_slot, (int64_t)i * m.nc, ab_finish24<UB_AHEAD>(m, ub_cur[half], left, i, 0));
                if (m.nc > 1) store_sample(p, m, pcm_slot, (int64_t)i * m.nc + 1, ab_finish24<UB_AHEAD>(m, ub_cur[half], right, i, 1));
            }
        }
        if (b < nch1) rf.commit();
    }
};

template <int NS>
__device__ __forceinline__ void ab_output_wave(const alac_decode_params& p, uint32_t pkt0, int lane, AbSharedT<NS>& sh, int nch0, int nch1) {
    AbOutBlock<NS> b0(p, pkt0, 0, lane, sh);
    if constexpr (NS == 16) {
        AbOutBlock<NS> b1(p, pkt0, 1, lane, sh);
        for (int b = 0; b <= nch0 + 1; b++) {
            wg_sync();
            b0.pass0_step(b, nch0);
            b1.pass0_step(b, nch0);
        }
        if (nch1 == 0) return;
        for (int b = 0; b <= nch1 + 1; b++) {
            wg_sync();
            b0.pass1_step(b, nch1);
            b1.pass1_step(b, nch1);
        }
    } else {
        for (int b = 0; b <= nch0 + 1; b++) {
            wg_sync();
            b0.pass0_step(b, nch0);
        }
        if (nch1 == 0) return;
        for (int b = 0; b <= nch1 + 1; b++) {
            wg_sync();
            b0.pass1_step(b, nch1);
        }
    }
}

// One FIR wave's two passes; the instantiation by what its streams need: rss > 23 anywhere -> the 32-bit multiply and the clamp.
template <int T, bool SPECIAL, int NS>
__device__ __forceinline__ void ab_fir_role(const alac_decode_params& p, uint32_t pkt0, int w, int lane, AbSharedT<NS>& sh, int nch0, int nch1,
                                            bool wide_rss) {
    if (T < 4 && !wide_rss) {
        for (int ph = 0; ph < (nch1 > 0 ? 2 : 1); ph++) ab_fir_wave<T, false, false, NS>(p, pkt0, w, lane, sh, ph, ph ? nch1 : nch0);
    } else {
        for (int ph = 0; ph < (nch1 > 0 ? 2 : 1); ph++) ab_fir_wave<T, true, SPECIAL, NS>(p, pkt0, w, lane, sh, ph, ph ? nch1 : nch0);
    }
}

// TSEL: which groups of 8 packets this body decodes, by the FIR layout they need --
//   1  first launch, 8-packet workgroups: orders 1..8 (one tap per lane); others are flagged for the second launch
//   0  first launch, dense arrangement (NS == 16): orders 1..16, one or two taps per lane per FIR wave; others flagged
//   2  second launch: the groups flagged 3 (some stream with 9..16 taps): two taps per lane
//   4  second launch: the groups flagged 1 (any order, the delta mode, order 0): four taps per lane
template <int TSEL, int NS = 8>
__device__ __forceinline__ void ab_kernel_body(const alac_decode_params& p, AbSharedT<NS>& sh) {
    static_assert((TSEL == 0) == (NS == 16), "the dense arrangement exists for the first launch only");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t pkt0 = blockIdx.x * (uint32_t)NS;
    // ab_flags[group]: 0 = decoded by the first launch; left for the second launch (alac_decode_ab32_kernel): 1 = to its
    // four-taps-per-lane code, 3 = to its two-taps-per-lane code; 2 / 4 = decoded there.
    // every wave reads all headers: pass lengths (uniform over the workgroup) and which FIR layout fits
    int n0 = 0, n1 = 0;
    bool bad = false, two_lane = false, wide_lane = false;
    {
        const uint32_t pk = pkt0 + (uint32_t)(lane & (NS - 1));
        const bool v = pk < p.n_packets;
        alacgpu_cfg_dev c;
        const Meta ma = parse_meta(p, pk, 0, v, c);
        const Meta mb = parse_meta(p, pk, 1, v, c);
        const bool ok = v && ma.status == 0;
        n0 = ok ? ma.n : 0;
        n1 = (ok && !ma.esc && ma.stereo) ? ma.n : 0;
        // one / two taps per lane take LPC orders 1..16; four taps per lane every order (0 and 31 as modes of the step)
        bad = TSEL < 4 && ok && !ma.esc && (ma.N < 1 || ma.N > 16 || (ma.stereo && (mb.N < 1 || mb.N > 16)) || (!ma.stereo && ma.predtype != 0));
        two_lane = ok && !ma.esc && (ma.N > 8 || (ma.stereo && mb.N > 8));
        wide_lane = ok && !ma.esc && ma.rss > 23;
        // the parking place needs two ints per sample in the slot (parse_meta turns anything else into a status)
        bad = bad || (n1 > 0 && (uint64_t)2 * (uint64_t)ma.n > p.slot_ints);
    }
    const bool fallback = __builtin_amdgcn_ballot_w64(bad) != 0;
    // some stream (of the block of 8 a FIR wave serves) has more than 8 taps / more than 23 bits
    const bool two0 = __builtin_amdgcn_ballot_w64(two_lane && (lane & (NS - 1)) < 8) != 0;
    const bool two1 = NS > 8 && __builtin_amdgcn_ballot_w64(two_lane && (lane & (NS - 1)) >= 8) != 0;
    const bool wrss0 = __builtin_amdgcn_ballot_w64(wide_lane && (lane & (NS - 1)) < 8) != 0;
    const bool wrss1 = NS > 8 && __builtin_amdgcn_ballot_w64(wide_lane && (lane & (NS - 1)) >= 8) != 0;
    if (TSEL <= 1) {
        // (8-packet arrangement: groups with 9..16 taps go to the second launch; the dense one keeps them)
        const bool hand_over = TSEL == 1 && two0 && !fallback;
        if (p.ab_flags && threadIdx.x == 0) {   // one flag per 8 packets (the second launch works in groups of 8)
            const uint32_t f0 = blockIdx.x * (uint32_t)(NS / 8);
            p.ab_flags[f0] = fallback ? 1u : hand_over ? 3u : 0u;
            if (NS > 8 && pkt0 + 8u < p.n_packets) p.ab_flags[f0 + 1] = fallback ? 1u : 0u;
        }
        if (hand_over || fallback) return;
    } else {
        if (threadIdx.x == 0) p.ab_flags[blockIdx.x] = TSEL == 2 ? 4u : 2u;
        // Nothing is launched behind this kernel.  What the four-taps body cannot take would be a two-channel packet without
        // room for parking in its slot -- parse_meta turns that into a status.  Should that ever change, fail loudly:
        if (fallback) {
            if (threadIdx.x < (unsigned)NS && pkt0 + threadIdx.x < p.n_packets) {
                p.status[pkt0 + threadIdx.x] = ALACGPU_ST_UNSUPPORTED_PARAMS_D;
                if (p.out_bytes) p.out_bytes[pkt0 + threadIdx.x] = 0;
                if (p.out_samples) p.out_samples[pkt0 + threadIdx.x] = 0;
            }
            return;
        }
    }
    const int nch0 = (__builtin_amdgcn_readfirstlane(wave_max(n0)) + AB_CHUNK - 1) / AB_CHUNK;
    const int nch1 = (__builtin_amdgcn_readfirstlane(wave_max(n1)) + AB_CHUNK - 1) / AB_CHUNK;
    // the zero column of the converted queue
    for (int t = threadIdx.x; t < 2 * AB_CHUNK; t += blockDim.x) xq_zero(sh.xq[t / AB_CHUNK][t % AB_CHUNK][NS]);
    DIAG_ONLY(if (p.dbg && threadIdx.x == 0) {
        p.dbg[8 * blockIdx.x + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
        p.dbg[8 * blockIdx.x + 0] = clock64();
    })
    // Who does what.  A heavy wave (entropy, FIR) should share its SIMD with an output wave at most: where two heavy waves of
    // different workgroups share one, those workgroups take 30 % longer, and the launch ends with its slowest workgroup.
    // Where the dispatcher puts the waves depends on what ran before (even on the kernel launched before this one: measured,
    // one XCD's worth of CUs ended up with clashing pairs and cfg2 took 1.05 instead of 0.81 ms), so the roles do not go
    // by wave index: every workgroup has one wave on each of the CU's four SIMDs (the 8-packet arrangement brings a fourth
    // wave for that, which leaves at once), takes its turn k on the CU from a counter, and puts entropy on SIMD k, output on
    // k + 1, FIR on k + 2 (the dense arrangement: its second FIR wave on k + 3).  Workgroups with consecutive turns so
    // never pair two heavy waves; four per CU load every SIMD alike.
    int role = wave;
    if (p.cu_arrivals) {
        const uint32_t hw = __builtin_amdgcn_s_getreg(63492);                       // HW_ID
        const uint32_t my_simd = (hw >> 4) & 3u;
        if (threadIdx.x == 0) {
            const uint32_t cu = ((__builtin_amdgcn_s_getreg(63508) & 7u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
            sh.ring_next[NS - 1] = atomicAdd(&p.cu_arrivals[cu], 1u);               // (ring_next / ring_on are free until the first pass)
        }
        if (lane == 0) sh.ring_on[wave] = my_simd;
        wg_sync();
        const uint32_t used = (1u << sh.ring_on[0]) | (1u << sh.ring_on[1]) | (1u << sh.ring_on[2]) | (1u << sh.ring_on[3]);
        if (used == 15u) role = (int)((my_simd - sh.ring_next[NS - 1]) & 3u);       // (else: not one wave per SIMD -- wave order)
    }
    // Orders above 16: small batches (one workgroup per CU at most: every heavy wave has a SIMD to itself and the launch is as
    // long as the FIR step's serial chain) split the 8 streams over TWO FIR waves with 16 lanes per stream and two taps per lane
    // (a step of 41 issue slots instead of 65); bigger batches, where the SIMDs are shared, take the ONE wave with 8 lanes per
    // stream and four taps per lane (65 slots per 8 streams instead of 2 x 41).  Measured: profiles/experiments/r3_fir_16_lanes.txt
    const bool two_fir = TSEL == 4 && gridDim.x <= (unsigned)ALAC_L16_MAX_GROUPS;
    if (NS == 8 && !two_fir && role == 3) return;     // the fourth wave was only there to claim the fourth SIMD
    wg_sync();
    if (role == 0) {
        __builtin_amdgcn_s_setprio(ALAC_ENTROPY_PRIO);
        ab_entropy_wave<NS>(p, pkt0, lane, sh, nch0, nch1);
    } else if (role == 1) {
        ab_output_wave<NS>(p, pkt0, lane, sh, nch0, nch1);
    } else {
        __builtin_amdgcn_s_setprio(1);   // above the output waves, below the entropy waves (8192 packets: 1.074 -> 1.048 ms, cfg3 3.33 -> 3.25)
        const int w = role - 2;
        const bool wr = w ? wrss1 : wrss0;
        if constexpr (TSEL == 4) {
            if (two_fir) {
                for (int ph = 0; ph < (nch1 > 0 ? 2 : 1); ph++) ab_fir_wave<2, true, true, NS, 16>(p, pkt0, w, lane, sh, ph, ph ? nch1 : nch0);
            } else {
                ab_fir_role<4, true, NS>(p, pkt0, w, lane, sh, nch0, nch1, true);
            }
        }
        else if constexpr (TSEL == 2) ab_fir_role<2, false, NS>(p, pkt0, w, lane, sh, nch0, nch1, wr);
        else if constexpr (TSEL == 1) ab_fir_role<1, false, NS>(p, pkt0, w, lane, sh, nch0, nch1, wr);
        else if (__builtin_expect(!(w ? two1 : two0), 1)) ab_fir_role<1, false, NS>(p, pkt0, w, lane, sh, nch0, nch1, wr);
        else ab_fir_role<2, false, NS>(p, pkt0, w, lane, sh, nch0, nch1, wr);
    }
}

}  // namespace

// One kernel entry per object file: the Makefile compiles this source five times (-DALAC_EMIT=1 .. 5), each kernel with
// the instruction-scheduler settings it measured best with (Makefile: SCHED_*).  Without ALAC_EMIT all of them are emitted.
#if !defined(ALAC_EMIT) || ALAC_EMIT == 1
// The main kernel for batches whose workgroups fit the chip four per CU (up to 10240 packets): 128 registers, nothing spilled.
extern "C" __global__ __launch_bounds__(256, 4) void alac_decode_ab_kernel(alac_decode_params p) {
    __shared__ __attribute__((aligned(1024))) AbSharedT<8> sh;
    ab_kernel_body<1>(p, sh);
}
#endif
#if !defined(ALAC_EMIT) || ALAC_EMIT == 5
// The same as alac_decode_ab_kernel, its object compiled with speculative units of 16 steps (-DALAC_SPEC_UNIT=16), for batches
// of up to 4096 packets: the entropy wave's straight-line blocks are twice as long and the code between units runs half as
// often, a failed unit costs twice as much: better where the launch is bound by one packet's chain, worse from 5120 packets
// on, where wasted instructions count.
extern "C" __global__ __launch_bounds__(256, 4) void alac_decode_ab_small_kernel(alac_decode_params p) {
    __shared__ __attribute__((aligned(1024))) AbSharedT<8> sh;
    ab_kernel_body<1>(p, sh);
}
#endif
#if !defined(ALAC_EMIT) || ALAC_EMIT == 4
// The same with 96 registers, for the batches in between (10241 .. 12288 packets): five workgroups per CU instead of four
// once a batch has more than fit at once.
extern "C" __global__ __launch_bounds__(256, 5) void alac_decode_ab5_kernel(alac_decode_params p) {
    __shared__ __attribute__((aligned(1024))) AbSharedT<8> sh;
    ab_kernel_body<1>(p, sh);
}
#endif
#if !defined(ALAC_EMIT) || ALAC_EMIT == 2
// The second launch: the groups the first one flagged -- two taps per lane (orders 9..16), four (anything else)
extern "C" __global__ __launch_bounds__(256) void alac_decode_ab32_kernel(alac_decode_params p) {
    __shared__ __attribute__((aligned(1024))) AbSharedT<8> sh;
    const uint32_t f = p.ab_flags ? p.ab_flags[blockIdx.x] : 1u;
    if (f == 1u) ab_kernel_body<4>(p, sh);
    else if (f == 3u) ab_kernel_body<2>(p, sh);
}
#endif
#if !defined(ALAC_EMIT) || ALAC_EMIT == 3
// The dense arrangement for big batches: 16 packets per 256-thread workgroup: one entropy wave for all 16 streams (4 lanes
// each), one output wave for all 16, two FIR waves of 8 streams -- one wave per SIMD, roles by SIMD and turn as above.  Same
// results, fewer instructions per sample; a step of its entropy wave takes as long as the 8-stream one's, so small
// (latency-bound) batches gain nothing.
extern "C" __global__ __launch_bounds__(256, 4) void alac_decode_ab_dense_kernel(alac_decode_params p) {
    __shared__ __attribute__((aligned(1024))) AbSharedT<16> sh;
    ab_kernel_body<0, 16>(p, sh);
}
#endif

// alac_device.h -- device-side building blocks shared by the ALAC decode kernels (gfx950 only).
// Bit reader, adaptive Golomb-Rice step, adaptive FIR step, header parse.  Reference line numbers are
// ALACDecoder/AlacFile.cs of teekay/ALAC.NET.
#ifndef ALAC_DEVICE_H
#define ALAC_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "alac_kernels.h"

namespace alacdev {


constexpr int RING_BYTES = 1024;          // per-row LDS ring
constexpr int RING_MASK = RING_BYTES - 1;
// a refill chunk is LPS lanes x 16 B, LPS = lanes cooperating on one stream
constexpr int BUFFER_SIZE = 16384;        // AlacFile.cs:28

// LDS hand-off between lanes of ONE wave: a wave's LDS operations execute in order, so only the
// compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// LDS hand-off between the waves of a workgroup.
__device__ __forceinline__ void wg_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- C# int semantics: wrapping add/sub/mul ---------------------------------------------------------
__device__ __forceinline__ int wadd(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int wsub(int a, int b) { return (int)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int wmul(int a, int b) { return (int)((uint32_t)a * (uint32_t)b); }

// ---- DPP helpers (row = 16 lanes) --------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp0(int v) {  // out-of-row / invalid source lanes read 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
constexpr int DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_ROW_SHL1 = 0x101, DPP_ROW_SHR1 = 0x111;

// ---- slow-path bit access straight from global memory (header, shift bytes, escape samples) ----
// `limit` is the packet's readable byte count from `base` (Meta::limit): bytes at or past it read as ZERO, exactly like
// the oracle's reader (and unlike the reference, which would see stale bytes of its reused 80 KiB buffer,
// AlacContext.cs:64,195).  The device allocation is readable up to blob_bytes rounded up to 16, base is 16-byte aligned.
__device__ __forceinline__ uint32_t load_be32(const uint8_t* base, int64_t byte_off, int64_t limit) {
    // byte_off is 4-aligned relative to a 16-aligned base
    if (byte_off < 0 || byte_off >= limit) return 0;
    uint32_t v = __builtin_bswap32(*reinterpret_cast<const uint32_t*>(base + byte_off));
    const int64_t nv = limit - byte_off;                       // valid bytes in this dword when < 4
    if (nv < 4) v &= ~(0xFFFFFFFFu >> (8 * (int)nv));
    return v;
}
// 16 bytes at a 16-aligned offset, zero past `limit` (raw little-endian dwords, before the byte swap)
__device__ __forceinline__ uint4 load16_clamped(const uint8_t* base, int64_t off, int64_t limit) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (off < limit) {
        v = *reinterpret_cast<const uint4*>(base + off);
        if (__builtin_expect(off + 16 > limit, 0)) {           // the packet ends inside these 16 bytes (once per stream)
            const int nv = (int)(limit - off);                 // 1..15 valid bytes
            uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int nb = nv - 4 * i;
                w[i] = nb >= 4 ? w[i] : (nb <= 0 ? 0u : (w[i] & ((1u << (8 * nb)) - 1u)));
            }
        }
    }
    return v;
}
// nbits in 1..32, MSB-first field starting at bit `bitpos`
__device__ __forceinline__ uint32_t peek_bits(const uint8_t* base, int64_t limit, uint32_t bitpos, int nbits) {
    int64_t d = (int64_t)(bitpos >> 5) * 4;
    uint32_t hi = load_be32(base, d, limit), lo = load_be32(base, d + 4, limit);
    uint64_t w = ((uint64_t)hi << 32) | lo;
    w <<= (bitpos & 31);
    return (uint32_t)(w >> 32) >> (32 - nbits);
}

// ---- LDS by integer address: lets the ring wrap be one v_and_or_b32 -----------------------------------
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
__device__ __forceinline__ uint32_t lds_addr(const uint32_t* p) { return (uint32_t)(uintptr_t)(const lds_u32_t*)p; }
__device__ __forceinline__ uint32_t lds_load(uint32_t a) { return *(const lds_u32_t*)(uintptr_t)a; }

// ---- Rice reader state (row-uniform) -------------------------------------------------------------
struct Rice {
    uint32_t w0, w1, w2;  // three consecutive big-endian dwords; w2 is the prefetched one
    uint32_t cur;         // bit cursor, NOT normalised: cur & 31 = unconsumed bits left in w0 (what v_alignbit takes),
                          // ~(cur >> 5) & 255 = ring dword of w2.  Consuming c bits is cur -= c (see rice_advance)
    uint32_t ra;          // LDS byte address of w2's dword inside the (1 KiB aligned) ring == rice_w2_addr(cur)
    uint32_t ra_sync;     // value of ra when `next` was last brought up to date (rice_sync)
    uint32_t next;        // byte offset (from the aligned packet base) of the dword after w2; lazily updated
    int hist;             // history            (AlacFile.cs:216)
    int signmod;          // signModifier       (:218)
    int zrun;             // zeros still to emit from the last run (:238-245)
    uint32_t nforce;      // ~0 when the lane may take the fast step, 0 when it needs the generic one
                          // (inside a zero run, or signModifier pending)
};

// The cursor that goes with `rem` unconsumed bits in w0 and w2 at LDS address ra.
__device__ __forceinline__ uint32_t rice_cursor(int rem, uint32_t ra) {
    return (uint32_t)rem - 32u - 8u * (ra & (uint32_t)RING_MASK);
}
// LDS address of w2 for a cursor (ring base is 1 KiB aligned): one v_lshrrev + one v_bitop3.
__device__ __forceinline__ uint32_t rice_w2_addr(uint32_t cur, uint32_t ring) {
    uint32_t a = (~(cur >> 3) & 0x3FCu) | ring;
    asm("" : "+v"(a));   // keep it one three-input bit op; callers compare whole addresses
    return a;
}
__device__ __forceinline__ uint32_t rice_window(const Rice& s) {
    return __builtin_amdgcn_alignbit(s.w0, s.w1, s.cur);
}
// Consumes c <= 32 bits: the window slides by at most one dword, exactly when w2's address changes.
__device__ __forceinline__ void rice_advance(Rice& s, int c, uint32_t ring) {
    s.cur -= (uint32_t)c;
    const uint32_t a2 = rice_w2_addr(s.cur, ring);
    const bool adv = a2 != s.ra;
    s.w0 = adv ? s.w1 : s.w0;
    s.w1 = adv ? s.w2 : s.w1;
    s.ra = a2;
    s.w2 = lds_load(a2);
}
// Brings s.next up to date.  Must be called at least once per RING_BYTES of consumption (callers do it
// every 16 samples, <= 118 bytes) and before ring_fill / rice_bitpos.
__device__ __forceinline__ void rice_sync(Rice& s) {
    s.next += (s.ra - s.ra_sync) & RING_MASK;
    s.ra_sync = s.ra;
}
__device__ __forceinline__ uint32_t rice_bitpos(const Rice& s) { return (s.next - 12u) * 8u + 32u - (s.cur & 31u); }

// One EntropyDecodeValue (AlacFile.cs:193-212).  m = ((1<<k)-1) & mask, escape_bits = rss or 16.
__device__ __forceinline__ uint32_t rice_symbol(Rice& s, int k, uint32_t m, int escape_bits, uint32_t ring) {
    uint32_t win = rice_window(s);
    uint32_t x = (uint32_t)__clz((int)~win);  // leading ones; 32 when win is all ones
    uint32_t v;
    if (x > 8) {                              // nine 1s: raw value follows (:198-202)
        rice_advance(s, 9, ring);
        win = rice_window(s);
        v = win >> (32 - escape_bits);
        rice_advance(s, escape_bits, ring);
    } else {
        uint32_t e = (win << (x + 1)) >> (32 - k);      // Readbits(k)            (:205)
        uint32_t big = e > 1 ? 1u : 0u;
        v = __umul24(x, m) + (big ? e - 1 : 0);         // (:206-208)
        rice_advance(s, (int)(x + k + big), ring);      // Unreadbits(1) when e <= 1 (:210)
    }
    return v;
}

struct RiceCfg {
    int kmod;
    uint32_t kmask;
    int hist_mult;
    int rss;
    // derived (rice_cfg_finish), for the plain speculative step:
    uint32_t keep_mult;    // 512 - hist_mult
    uint32_t keep_bias;    // 1536 * hist_mult + 511
    int cfloor;            // the smallest clz(history + 1536) the plain step is exact for
};

// The "compressed blocks of 0" branch of EntropyRiceDecode (AlacFile.cs:231-249), taken right after a
// value whose updated history fell below 128 (and a sample remains).  s.hist holds that history.
__device__ __forceinline__ void rice_run_part(Rice& s, const RiceCfg& c, int sample_idx, int* flags,
                                              uint32_t ring) {
    int h = s.hist;
    if (h < 0) { *flags |= 2; h = 0; }
    s.signmod = 1;                                                            // :233
    int k2 = (h == 0 ? 40 : __clz(h)) + ((h + 16) >> 6) - 24;                // :234 (clz(0) == 40 quirk)
    uint32_t bs = rice_symbol(s, k2, ((1u << (k2 & 31)) - 1u) & c.kmask, 16, ring);  // :236
    if ((uint32_t)sample_idx + bs > (uint32_t)(BUFFER_SIZE - 1)) *flags |= 1;        // :242 would throw
    s.zrun = bs > 0x7FFFFFFFu ? 0x7FFFFFFF : (int)bs;
    if (bs > 0xFFFF) s.signmod = 0;                                           // :246
    s.hist = 0;                                                               // :248
    s.nforce = 0;
}

// One output residual of EntropyRiceDecode (AlacFile.cs:219-251), every case.  `remaining` =
// outputSize-1-outputCount.  Sets *flags bit0 when a zero run would leave the reference's 16384-entry
// scratch, bit1 when history went negative.
__device__ __forceinline__ int rice_step(Rice& s, const RiceCfg& c, int remaining, int sample_idx, int* flags,
                                         uint32_t ring) {
    int r = 0;
    if (s.zrun > 0) {
        s.zrun--;
    } else {
        int t = (s.hist >> 9) + 3;
        int k = 31 - __clz(t);
        k = k < c.kmod ? k : c.kmod;                                             // :221-222
        uint32_t dv = rice_symbol(s, k, (1u << k) - 1u, c.rss, ring) + (uint32_t)s.signmod;  // :224
        s.signmod = 0;
        r = (int)(dv >> 1) ^ -(int)(dv & 1u);                                    // :225-226 (dv >= 0)
        int h = s.hist;
        h = (int)dv > 0xFFFF ? 0xFFFF : wsub(wadd(h, wmul((int)dv, c.hist_mult)), wmul(h, c.hist_mult) >> 9);  // :229
        s.hist = h;
        if (h < 128 && remaining > 0) rice_run_part(s, c, sample_idx, flags, ring);  // :231
    }
    s.nforce = (s.zrun > 0 || s.signmod != 0) ? 0u : 0xFFFFFFFFu;
    return r;
}

// Speculative fast step: the common case of rice_step (no pending zero run / signModifier, unary prefix
// <= 8, updated history >= 128) as straight-line code with no branch and no exec masking.  It does NOT
// check those conditions: it records them in xmax / hmin so that the caller can validate a whole unit of
// steps afterwards and, if any lane left the common case, restore its snapshot and redo the unit with
// rice_step.  A lane that has left the common case keeps running on garbage; that is harmless (LDS ring
// reads are address-masked, nothing else is touched).
// WANT_R: return the residual (else 0).  RAW: return the unsigned code value dv instead of the signed residual
// (dv >> 1) ^ -(dv & 1) (:225-226) -- the output wave, which has cycles to spare, does that conversion (xq_from_code).
// What the plain step leaves out, and how the caller finds out (rice_plain_ok) that nothing was missed:
//  * s.hist holds H = history + 1536 while the plain steps run (the caller adds and removes the bias): k = 22 - clz(H)
//    without the addition, and history - ((history * mult) >> 9) is one 24-bit multiply-add and a shift (c.keep_mult = 512 -
//    mult, c.keep_bias = 1536 * mult + 511) -- exact while H < 2^23 (the sum stays below 2^32) and history * mult < 2^31
//    (the reference's product does not wrap);
//  * k is not capped at kmod (:222), the history clamp for values above 0xFFFF (:229) is not applied: the step tracks the
//    smallest clz(H) of the unit in cmin instead.  cmin >= c.cfloor says that no k of the unit was above kmod (and that H
//    stayed inside the exact range above); with the unit's largest prefix xmax, (xmax + 1) << (22 - cmin) <= 65536 says that no
//    value was above 0xFFFF (a value is below (x + 1) << k).  A unit that fails either test goes to the escape tier, whose
//    step caps and clamps.
template <bool WANT_R, bool RAW = false>
__device__ __forceinline__ int rice_spec_step(Rice& s, const RiceCfg& c, uint32_t ring, uint32_t& xmax,
                                              int& hmin, uint32_t& cmin) {
    const uint32_t win = rice_window(s);
    const uint32_t x = (uint32_t)__builtin_clz(~win | 0x00400000u);          // leading ones, capped at 9 (:196)
    xmax = max(xmax, x);
    const uint32_t H = (uint32_t)s.hist;                                     // history + 1536
    // (hist >> 9) + 3 == (hist + 1536) >> 9 for hist >= 0, so 31 - clz((hist >> 9) + 3) = 22 - clz(H)    (:221)
    const uint32_t lz = (uint32_t)__builtin_clz(H);
    cmin = min(cmin, lz);
    int k = 22 - (int)lz;
    asm("" : "+v"(k));                  // (keeps x + k one sum used twice: the compiler otherwise spreads 22 - lz over both uses)
    uint32_t xk = x + (uint32_t)k;
    asm("" : "+v"(xk));
    const uint32_t e = __builtin_amdgcn_ubfe(win, 31u - xk, (uint32_t)k);          // Readbits(k) (:205)
    const uint32_t m = __builtin_amdgcn_ubfe(0xFFFFFFFFu, 0u, (uint32_t)k);        // (1 << k) - 1
    const uint32_t v = __umul24(x, m) + (e > 1u ? e - 1u : 0u);                    // :206-208
    const uint32_t cur2 = s.cur - xk - (e > 1u ? 1u : 0u);                   // bits used: x+1+k, minus the un-read one (:210)
    int r = 0;
    if (WANT_R) r = RAW ? (int)v : (int)(v >> 1) ^ -(int)(v & 1u);           // :225-226
    // H - ((history * mult) >> 9) = ceil((512 H - history * mult) / 512) = (H * (512 - mult) + 1536 * mult + 511) >> 9
    const uint32_t kept = (__umul24(H, c.keep_mult) + c.keep_bias) >> 9;
    const int hn = (int)(__umul24(v, (uint32_t)c.hist_mult) + kept);         // :229 without the clamp, still biased
    hmin = min(hmin, hn);
    const uint32_t a2 = rice_w2_addr(cur2, ring);
    const bool adv = a2 != s.ra;
    s.cur = cur2;
    s.w0 = adv ? s.w1 : s.w0;
    s.w1 = adv ? s.w2 : s.w1;
    s.ra = a2;
    s.w2 = lds_load(a2);
    s.hist = hn;
    return r;
}
constexpr int RICE_PLAIN_BIAS = 1536;
// The verdict on a unit of narrow plain steps, per lane, escape codes aside: no value above 0xFFFF, no k above kmod, history
// inside the exact range (see rice_spec_step).
__device__ __forceinline__ bool rice_narrow_ok(const RiceCfg& c, uint32_t xmax, uint32_t cmin) {
    const uint32_t kmax = (22u - cmin) & 31u;
    return cmin >= (uint32_t)c.cfloor && cmin <= 22u && ((xmax + 1u) << kmax) <= 0x10000u;
}
// The derived members of RiceCfg.
__device__ __forceinline__ void rice_cfg_finish(RiceCfg& c) {
    // hist_mult = ricemodifier (3 bits) * (rice_history_mult (a byte) / 4) <= 441 < 512
    c.keep_mult = 512u - (uint32_t)c.hist_mult;
    c.keep_bias = (uint32_t)RICE_PLAIN_BIAS * (uint32_t)c.hist_mult + 511u;
    const int mbits = 32 - __builtin_clz((uint32_t)c.hist_mult | 1u);        // hist_mult < 2^mbits
    c.cfloor = max(max(9, mbits + 1), 22 - c.kmod);
}
// The plain step for every history and every kmod (streams the narrow step's arithmetic does not hold: k at its cap, history
// + 1536 >= 2^23 -- full-scale noise): k capped (:222), the decay term by a full 32-bit multiply.  It leaves out the history
// clamp for values above 0xFFFF (:229; two instructions) and tracks the largest value in vmax instead; the caller sends a unit
// with vmax > 0xFFFF to the escape tier, whose step does clamp.
template <bool WANT_R, bool RAW = false>
__device__ __forceinline__ int rice_spec_step_wide(Rice& s, const RiceCfg& c, uint32_t ring, uint32_t& xmax,
                                              int& hmin, uint32_t& vmax) {
    const uint32_t win = rice_window(s);
    const uint32_t x = (uint32_t)__builtin_clz(~win | 0x00400000u);          // leading ones, capped at 9 (:196)
    xmax = max(xmax, x);
    // (hist >> 9) + 3 == (hist + 1536) >> 9 for hist >= 0, so k = min(31 - clz(..), kmod) = min(22 - clz(hist + 1536), kmod)
    const int k = min(22 - __builtin_clz((uint32_t)(s.hist + 1536)), c.kmod);      // :221-222
    const uint32_t e = __builtin_amdgcn_ubfe(win, (uint32_t)(31 - k) - x, (uint32_t)k);  // Readbits(k) (:205)
    const uint32_t m = __builtin_amdgcn_ubfe(0xFFFFFFFFu, 0u, (uint32_t)k);        // (1 << k) - 1
    const uint32_t v = __umul24(x, m) + (e > 1u ? e - 1u : 0u);                    // :206-208
    vmax = max(vmax, v);
    const uint32_t cur2 = s.cur - (x + (uint32_t)k) - (e > 1u ? 1u : 0u);    // bits used: x+1+k, minus the un-read one (:210)
    int r = 0;
    if (WANT_R) r = RAW ? (int)v : (int)(v >> 1) ^ -(int)(v & 1u);           // :225-226
    const int h = s.hist;
    const int hn = (int)(__umul24(v, (uint32_t)c.hist_mult) + (uint32_t)h) - (wmul(h, c.hist_mult) >> 9);   // :229 without the clamp
    hmin = min(hmin, hn);
    const uint32_t a2 = rice_w2_addr(cur2, ring);
    const bool adv = a2 != s.ra;
    s.cur = cur2;
    s.w0 = adv ? s.w1 : s.w0;
    s.w1 = adv ? s.w2 : s.w1;
    s.ra = a2;
    s.w2 = lds_load(a2);
    s.hist = hn;
    return r;
}
// rice_spec_step plus escape codes (nine 1s + rss raw bits, AlacFile.cs:198-202), for rss <= 23: the raw value then lies
// inside the 32-bit window (9 + rss <= 32), and a step still consumes at most 32 bits -- the window slides by at most one
// dword, so this is the plain step plus four instructions (the escape-capable rice_spec_step_full costs twenty more).
// Loud / noisy 16-bit content lives here.
template <bool WANT_R, bool RAW = false>
__device__ __forceinline__ int rice_spec_step_esc(Rice& s, const RiceCfg& c, uint32_t ring, uint32_t& xmax,
                                                  int& hmin) {
    const uint32_t win = rice_window(s);
    const uint32_t x = (uint32_t)__builtin_clz(~win | 0x00400000u);
    const bool esc = x > 8u;
    xmax = max(xmax, x);
    const int k = min(22 - __builtin_clz((uint32_t)(s.hist + 1536)), c.kmod);
    const uint32_t e = __builtin_amdgcn_ubfe(win, (uint32_t)(31 - k) - x, (uint32_t)k);
    const uint32_t m = __builtin_amdgcn_ubfe(0xFFFFFFFFu, 0u, (uint32_t)k);
    const uint32_t vn = __umul24(x, m) + (e > 1u ? e - 1u : 0u);
    const uint32_t raw = __builtin_amdgcn_ubfe(win, (uint32_t)(23 - c.rss), (uint32_t)c.rss);   // bits 9 .. 9+rss of the window
    const uint32_t v = esc ? raw : vn;
    const uint32_t used = esc ? (uint32_t)(9 + c.rss) : x + (uint32_t)k + (e > 1u ? 1u : 0u);
    const uint32_t cur2 = s.cur - used;
    int r = 0;
    if (WANT_R) r = RAW ? (int)v : (int)(v >> 1) ^ -(int)(v & 1u);
    const int h = s.hist;
    int hx = (int)(__umul24(v, (uint32_t)c.hist_mult) + (uint32_t)h) - (wmul(h, c.hist_mult) >> 9);
    asm volatile("" : "+v"(hx));
    const int hn = (int)v > 0xFFFF ? 0xFFFF : hx;
    hmin = min(hmin, hn);
    const uint32_t a2 = rice_w2_addr(cur2, ring);
    const bool adv = a2 != s.ra;
    s.cur = cur2;
    s.w0 = adv ? s.w1 : s.w0;
    s.w1 = adv ? s.w2 : s.w1;
    s.ra = a2;
    s.w2 = lds_load(a2);
    s.hist = hn;
    return r;
}
// Same, for units in which some lane starts inside a zero run or with signModifier pending (digital
// silence): such a lane emits 0 without touching the bitstream while zrun > 0, and adds signModifier to
// its next value.  Still straight-line; only a NEW run symbol (history < 128 after a value) or an escape
// code sends the unit to rice_step.
template <bool WANT_R, bool RAW = false>
__device__ __forceinline__ int rice_spec_step_z(Rice& s, const RiceCfg& c, uint32_t ring, uint32_t& xmax,
                                                int& hmin) {
    const bool inrun = s.zrun > 0;
    const uint32_t win = rice_window(s);
    const uint32_t x = (uint32_t)__builtin_clz(~win | 0x00400000u);
    xmax = max(xmax, inrun ? 0u : x);
    const int k = min(22 - __builtin_clz((uint32_t)(s.hist + 1536)), c.kmod);
    const uint32_t e = __builtin_amdgcn_ubfe(win, (uint32_t)(31 - k) - x, (uint32_t)k);
    const uint32_t m = __builtin_amdgcn_ubfe(0xFFFFFFFFu, 0u, (uint32_t)k);
    const uint32_t v = __umul24(x, m) + (e > 1u ? e - 1u : 0u) + (uint32_t)s.signmod;   // :224
    const uint32_t cur2 = inrun ? s.cur : s.cur - (x + (uint32_t)k) - (e > 1u ? 1u : 0u);
    int r = 0;
    if (WANT_R) r = inrun ? 0 : (RAW ? (int)v : (int)(v >> 1) ^ -(int)(v & 1u));
    const int h = s.hist;
    int hx = (int)(__umul24(v, (uint32_t)c.hist_mult) + (uint32_t)h) - (wmul(h, c.hist_mult) >> 9);
    asm volatile("" : "+v"(hx));
    const int hv = (int)v > 0xFFFF ? 0xFFFF : hx;
    hmin = min(hmin, inrun ? 0x7FFFFFFF : hv);
    const uint32_t a2 = rice_w2_addr(cur2, ring);
    const bool adv = a2 != s.ra;
    s.cur = cur2;
    s.w0 = adv ? s.w1 : s.w0;
    s.w1 = adv ? s.w2 : s.w1;
    s.ra = a2;
    s.w2 = lds_load(a2);
    s.hist = inrun ? h : hv;
    s.signmod = inrun ? s.signmod : 0;
    s.zrun -= inrun ? 1 : 0;
    return r;
}
// Full-featured speculative step: additionally handles, still as straight-line code,
//   * lanes inside a zero run / with signModifier pending (digital silence): emit 0 without touching the
//     bitstream while zrun > 0, add signModifier to the next value;
//   * escape codes (nine 1s + rss raw bits, AlacFile.cs:198-202): loud / noisy content, where the history
//     clamp (:229) keeps k small and escapes are frequent.  An escape consumes up to 9 + 25 bits, so the
//     window slides by 0, 1 or 2 dwords per step and a fourth dword (w3) is kept prefetched.
// Only a NEW run symbol (history < 128 after a value) sends the unit to rice_step.
template <bool WANT_R, bool RAW = false>
__device__ __forceinline__ int rice_spec_step_full(Rice& s, uint32_t& w3, const RiceCfg& c, uint32_t ring, uint32_t& xmax,
                                                   int& hmin) {
    const bool inrun = s.zrun > 0;
    const uint32_t win = rice_window(s);
    const uint32_t win2 = __builtin_amdgcn_alignbit(s.w1, s.w2, s.cur);      // the 32 bits after `win`
    const uint32_t x = (uint32_t)__builtin_clz(~win | 0x00400000u);
    const bool esc = x > 8u;
    xmax = max(xmax, inrun ? 0u : x);
    const int k = min(22 - __builtin_clz((uint32_t)(s.hist + 1536)), c.kmod);
    const uint32_t e = __builtin_amdgcn_ubfe(win, (uint32_t)(31 - k) - x, (uint32_t)k);
    const uint32_t m = __builtin_amdgcn_ubfe(0xFFFFFFFFu, 0u, (uint32_t)k);
    const uint32_t vn = __umul24(x, m) + (e > 1u ? e - 1u : 0u);
    const uint32_t raw = __builtin_amdgcn_alignbit(win, win2, 23) >> (32 - c.rss);   // bits 9 .. 9+rss of the stream
    const uint32_t v = (esc ? raw : vn) + (uint32_t)s.signmod;                        // :224
    const int used = esc ? 9 + c.rss : (int)(x + (uint32_t)k) + (e > 1u ? 1 : 0);
    const uint32_t cur2 = inrun ? s.cur : s.cur - (uint32_t)used;             // up to 34 bits: w2 moves by 0, 1 or 2 dwords
    int r = 0;
    if (WANT_R) r = inrun ? 0 : (RAW ? (int)v : (int)(v >> 1) ^ -(int)(v & 1u));
    const int h = s.hist;
    int hx = (int)(__umul24(v, (uint32_t)c.hist_mult) + (uint32_t)h) - (wmul(h, c.hist_mult) >> 9);
    asm volatile("" : "+v"(hx));
    const int hv = (int)v > 0xFFFF ? 0xFFFF : hx;                             // :229 (hx is unused garbage when v is huge)
    hmin = min(hmin, inrun ? 0x7FFFFFFF : hv);
    const uint32_t na = rice_w2_addr(cur2, ring);
    const bool a1 = na != s.ra;                                               // slid by at least one dword
    const bool a2 = na == (((s.ra + 8u) & RING_MASK) | ring);                 // slid by two
    s.cur = cur2;
    const uint32_t n0 = a2 ? s.w2 : (a1 ? s.w1 : s.w0);
    const uint32_t n1 = a2 ? w3 : (a1 ? s.w2 : s.w1);
    s.w0 = n0;
    s.w1 = n1;
    s.ra = na;
    s.w2 = lds_load(na);
    w3 = lds_load(((na + 4u) & RING_MASK) | ring);
    s.hist = inrun ? h : hv;
    s.signmod = inrun ? s.signmod : 0;
    s.zrun -= inrun ? 1 : 0;
    return r;
}
// rice_spec_step_esc for any rss (24-bit streams: the raw value of an escape is up to 25 bits and reaches into the next
// window; a step consumes up to 34 bits, so the window slides by 0, 1 or 2 dwords and w3 is kept prefetched, as in
// rice_spec_step_full) -- but without that step's zero-run / signModifier selects.
template <bool WANT_R, bool RAW = false>
__device__ __forceinline__ int rice_spec_step_esc_wide(Rice& s, uint32_t& w3, const RiceCfg& c, uint32_t ring, uint32_t& xmax,
                                                       int& hmin) {
    const uint32_t win = rice_window(s);
    const uint32_t win2 = __builtin_amdgcn_alignbit(s.w1, s.w2, s.cur);
    const uint32_t x = (uint32_t)__builtin_clz(~win | 0x00400000u);
    const bool esc = x > 8u;
    xmax = max(xmax, x);
    const int k = min(22 - __builtin_clz((uint32_t)(s.hist + 1536)), c.kmod);
    const uint32_t e = __builtin_amdgcn_ubfe(win, (uint32_t)(31 - k) - x, (uint32_t)k);
    const uint32_t m = __builtin_amdgcn_ubfe(0xFFFFFFFFu, 0u, (uint32_t)k);
    const uint32_t vn = __umul24(x, m) + (e > 1u ? e - 1u : 0u);
    const uint32_t raw = __builtin_amdgcn_alignbit(win, win2, 23) >> (32 - c.rss);
    const uint32_t v = esc ? raw : vn;
    const uint32_t used = esc ? (uint32_t)(9 + c.rss) : x + (uint32_t)k + (e > 1u ? 1u : 0u);
    const uint32_t cur2 = s.cur - used;
    int r = 0;
    if (WANT_R) r = RAW ? (int)v : (int)(v >> 1) ^ -(int)(v & 1u);
    const int h = s.hist;
    int hx = (int)(__umul24(v, (uint32_t)c.hist_mult) + (uint32_t)h) - (wmul(h, c.hist_mult) >> 9);
    asm volatile("" : "+v"(hx));
    const int hn = (int)v > 0xFFFF ? 0xFFFF : hx;
    hmin = min(hmin, hn);
    const uint32_t na = rice_w2_addr(cur2, ring);
    const bool a1 = na != s.ra;
    const bool a2 = na == (((s.ra + 8u) & RING_MASK) | ring);
    s.cur = cur2;
    const uint32_t n0 = a2 ? s.w2 : (a1 ? s.w1 : s.w0);
    const uint32_t n1 = a2 ? w3 : (a1 ? s.w2 : s.w1);
    s.w0 = n0;
    s.w1 = n1;
    s.ra = na;
    s.w2 = lds_load(na);
    w3 = lds_load(((na + 4u) & RING_MASK) | ring);
    s.hist = hn;
    return r;
}
#ifndef ALAC_SPEC_UNIT
#define ALAC_SPEC_UNIT 8
#endif
constexpr int SPEC_UNIT = ALAC_SPEC_UNIT;   // steps per speculative unit

// ---- per-row LDS ring ------------------------------------------------------------------------------
// Tops the ring up with 256-byte chunks while there is room in front of the oldest live dword.
template <int LPS>
__device__ __forceinline__ void ring_fill(uint32_t* ring, uint32_t& filled, uint32_t next, const uint8_t* base,
                                          int64_t limit, int l, bool enable) {
    constexpr uint32_t FILL_CHUNK = LPS * 16;
    while (true) {
        bool need = enable && (filled + FILL_CHUNK <= (next - 12u) + RING_BYTES);
        if (!__builtin_amdgcn_ballot_w64(need)) break;
        if (need) {
            int64_t off = (int64_t)filled + l * 16;
            const uint4 v = load16_clamped(base, off, limit);
            uint4 o = make_uint4(__builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z),
                                 __builtin_bswap32(v.w));
            *reinterpret_cast<uint4*>(&ring[(off & RING_MASK) >> 2]) = o;
            filled += FILL_CHUNK;
        }
    }
}

template <int LPS>
__device__ __forceinline__ void rice_init(Rice& s, uint32_t& filled, uint32_t startbit, int init_hist, uint32_t* ring,
                                          const uint8_t* base, int64_t limit, int l, bool enable) {
    constexpr uint32_t FILL_CHUNK = LPS * 16;
    uint32_t p = startbit - 1u;   // startbit >= 23 always
    uint32_t d = (p >> 5) * 4u;   // byte offset of the dword holding bit startbit-1
    s.next = d + 12u;
    s.hist = init_hist;
    s.signmod = 0;
    s.zrun = 0;
    s.nforce = 0xFFFFFFFFu;
    filled = d & ~(uint32_t)(FILL_CHUNK - 1);
    wave_sync();
    ring_fill<LPS>(ring, filled, s.next, base, limit, l, enable);
    wave_sync();
    s.w0 = ring[((d) & RING_MASK) >> 2];
    s.w1 = ring[((d + 4u) & RING_MASK) >> 2];
    s.w2 = ring[((d + 8u) & RING_MASK) >> 2];
    s.ra = lds_addr(ring) | ((d + 8u) & RING_MASK);
    s.ra_sync = s.ra;
    s.cur = rice_cursor(31 - (int)(p & 31u), s.ra);
}

// ---- adaptive FIR (PredictorDecompressFirAdapt, AlacFile.cs:256-336): the "blocked" layout --------------------------
// 8 lanes per stream for EVERY order: in a row of 16 lanes the two parities hold the same channel of two different packets
// (lane = 2 jl + par), so one wave serves 8 streams; lane jl of a stream holds the T consecutive taps T jl .. T jl + T - 1 in
// T registers (T = 1: orders up to 8, T = 2: up to 16, T = 4: up to 32).  tap t = history value out[i-1-t], coefficient t.
//   * dot product: T multiply-adds inside the lane, then three DPP stages (quad_perm[2,3,0,1], row_ror:4, row_ror:8);
//   * sign-LMS early exit (:312-332; taps are visited from N-1 down to 0 while the running error keeps its sign): with
//     E = |err| and c_t = the magnitude tap t takes off it, tap t is visited iff E > sum of c_t' over t' > t -- a suffix sum:
//     T - 1 adds inside the lane, ONE three-stage DPP scan of the lane totals (row_shl:2/4/8, zero fill), T adds back;
//   * the history moves one tap per sample: inside a lane that is a renaming of registers (the unrolled steps rotate which
//     physical register plays tap T jl + r; PH = step number mod T), between lanes one row_shr:2 on the register that held
//     the lane's last tap, which lanes 0 / 1 of the row fill with the new sample;
//   * history values are kept BIASED by B = 2^(rss-1) (out_u = out + B, in [0, 2^rss) for every truncated sample): the
//     difference of two values is unchanged, the sign extension (:309-310) becomes one v_bfe_u32 of a sum that carries the
//     bias in through the base, and |hist - base| + rounding term is ONE v_sad_u32 (three instructions on signed values).
//     Only a first sample (:260, copied untruncated) or an order-0 stream can hold a value outside [0, 2^rss); the fast
//     step therefore runs from the second chunk on, when the first sample has left every window (N <= 30 < 32), and
//     order-0 streams have no taps (weight 0).  The masked step below works on signed differences and has no such condition.
// What a step needs from the residual arrives pre-digested (XQ; the output wave converts the entropy wave's code values
// between two barriers): the FIR wave is the one the common workgroup waits for, and every instruction of its step is
// issued once per sample for 8 streams.
struct XQ {          // one residual, as the FIR step wants it (16 bytes, one ds_read_b128)
    int err;         // the residual                                        (AlacFile.cs:225-226)
    int rq;          // err < 0 ? (1 << q) - 1 : 0    ((-a) >> q == -((a + 2^q - 1) >> q))
    uint32_t mag;    // |err|
    int sgn;         // err < 0 ? -1 : +1
};
struct XQ8 {         // the short form (the dense arrangement's LDS budget: four 16-packet workgroups per CU): the step
    int err;         // derives rq and the sign itself, three more instructions
    uint32_t mag;
};
__device__ __forceinline__ void xq_from_code(XQ& x, uint32_t dv, int qmask) {
    const int s = -(int)(dv & 1u);
    const int hq = (int)(dv >> 1);
    x.err = hq ^ s;                                   // :225-226
    x.rq = s & qmask;
    x.mag = (uint32_t)(hq - s);                       // (dv + 1) >> 1
    x.sgn = s | 1;
}
__device__ __forceinline__ void xq_from_code(XQ8& x, uint32_t dv, int) {
    const int s = -(int)(dv & 1u);
    const int hq = (int)(dv >> 1);
    x.err = hq ^ s;
    x.mag = (uint32_t)(hq - s);
}
__device__ __forceinline__ void xq_zero(XQ& x) { x.err = 0; x.rq = 0; x.mag = 0; x.sgn = 0; }
__device__ __forceinline__ void xq_zero(XQ8& x) { x.err = 0; x.mag = 0; }

// low 24 bits of a and b, signed, times each other plus c: exact mod 2^32 (one v_mad_i32_i24; a builtin rather than inline
// assembly, so that the compiler may speculate it: around inline assembly it builds a branch)
__device__ __forceinline__ int mad_i24(int a, int b, int c) { return wadd(__mul24(a, b), c); }
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t c) {   // |a - b| + c, unsigned
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int med3_i32(int a, int b, int c) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int xad_u32(int a, int b, int c) {   // (a ^ b) + c
    int r;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

constexpr int DPP_ROW_SHL_2 = 0x102, DPP_ROW_SHL_4 = 0x104, DPP_ROW_SHL_8 = 0x108, DPP_ROW_SHR2 = 0x112;
constexpr int DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;

template <int T>
struct FirB {
    int h[T];          // biased history: tap T jl + r (logical register r; see PH in firb_step)
    int c[T];          // predictorCoefTable[T jl + r] (0 past the stream's order)
    uint32_t w[T];     // N - t for tap t < N, else 0   (:329)
    int tlo[T], thi[T];// -1 / +1 on tap lanes, 0 / 0 elsewhere: bounds of the sign() median
    int base;          // biased out[i-1-N]; delta mode: out[i-1]
    int prev;          // biased out[i-1]: kept by the masked step only
    int rnd0;          // 1 << (q-1) in lane jl == 0 of a stream in the general mode, 0 elsewhere: enters the sum once
    int q, rss, qmask, bias;
    int N;             // 0 for a stream that is switched off
    int bpaddr;        // ds_bpermute byte address of the lane that holds tap N-1
    int bsel;          // ... and its register there, (N-1) % T
    bool n0, delta;    // order 0 (out = err, :260-267) / order 31 (out = sx(prev + err), :268-282)
};

// All-reduce over the L lanes of a stream / inclusive suffix sum over them (lane jl gets the sum over lanes >= jl) / the DPP
// control that moves a value one lane of the stream up.  L = 8: the lanes of a stream are every other lane of a row of 16
// (two streams per row); L = 16: a row is a stream (orders above 16: two FIR waves of four streams share the 8 streams).
template <int L>
__device__ __forceinline__ int firb_allreduce(int p) {
    if (L == 16) p = wadd(p, dpp0<DPP_QUAD_1032>(p));
    p = wadd(p, dpp0<DPP_QUAD_2301>(p));
    p = wadd(p, __builtin_amdgcn_update_dpp(0, p, DPP_ROW_ROR4, 0xF, 0xF, false));
    p = wadd(p, __builtin_amdgcn_update_dpp(0, p, DPP_ROW_ROR8, 0xF, 0xF, false));
    asm("" : "+v"(p));   // keep the last stage a one-instruction v_add_u32_dpp (merged into a v_add3 it needs a v_mov_dpp first)
    return p;
}
template <int L>
__device__ __forceinline__ uint32_t firb_suffix(uint32_t v) {
    if (L == 16) v += (uint32_t)dpp0<DPP_ROW_SHL1>((int)v);
    v += (uint32_t)dpp0<DPP_ROW_SHL_2>((int)v);
    v += (uint32_t)dpp0<DPP_ROW_SHL_4>((int)v);
    v += (uint32_t)dpp0<DPP_ROW_SHL_8>((int)v);
    return v;
}

// Steady state (every stream of the wave switched on and past its warm-up, i > N, no first sample left in a window).
// WIDE: some stream of the wave has rss > 23 (the 24-bit multiply-add would drop bits of hist - base; the clamp keeps the
// suffix sums from wrapping).  SPECIAL: some stream is of order 0 or 31.  PH: step number mod T (register rotation).
template <int T, bool WIDE, bool SPECIAL, int PH, int L, typename E>
__device__ __forceinline__ void firb_step(FirB<T>& f, const E e) {
    constexpr bool SHORT = sizeof(E) == sizeof(XQ8);
    XQ x;
    int smask = 0;
    if constexpr (SHORT) {
        x.err = e.err;
        x.mag = e.mag;
        smask = e.err >> 31;
        x.rq = smask & f.qmask;
        x.sgn = 0;
    } else {
        x = e;
    }
    constexpr int P0 = ((0 - PH) % T + T) % T, PL = ((T - 1 - PH) % T + T) % T;
    int src = f.h[P0];
#pragma unroll
    for (int r = 1; r < T; r++) src = (f.bsel == r) ? f.h[((r - PH) % T + T) % T] : src;
    const int nb = __builtin_amdgcn_ds_bpermute(f.bpaddr, src);     // tap N-1 = the next step's base; needed last
    int d[T], sd[T];
    uint32_t cc[T];
    int p = 0;
#pragma unroll
    for (int r = 0; r < T; r++) {
        const int hr = f.h[((r - PH) % T + T) % T];
        d[r] = wsub(hr, f.base);                                              // :303
        if (!WIDE) p = mad_i24(d[r], f.c[r], r == 0 ? f.rnd0 : p);
        else p = wadd(wmul(d[r], f.c[r]), r == 0 ? f.rnd0 : p);
        const uint32_t aq = sad_u32((uint32_t)hr, (uint32_t)f.base, (uint32_t)x.rq) >> f.q;
        cc[r] = aq * f.w[r];                                                  // what the tap takes off |err| (:329); < 2^31
        if (WIDE && T == 1) cc[r] = min(cc[r], 1u << 26);                     // keeps the scan from wrapping; decisions unchanged: |err| < 2^26
        sd[r] = med3_i32(d[r], f.tlo[r], f.thi[r]);                           // sign(hist - base) on tap lanes
    }
    const int sum = firb_allreduce<L>(p);
    const int xu = wadd(wadd(sum >> f.q, f.base), x.err);                     // :306-308 (biased through the base)
    int out = (int)__builtin_amdgcn_ubfe((uint32_t)xu, 0u, (uint32_t)f.rss); // :309-310
    if (SPECIAL) out = f.n0 ? wadd(x.err, f.bias) : out;                      // order 0 copies (:260-267)
    bool visit[T];
    if (T == 1) {
        visit[0] = x.mag + cc[0] > firb_suffix<L>(cc[0]);
    } else if (!WIDE) {
        uint32_t tot = cc[0];
#pragma unroll
        for (int r = 1; r < T; r++) tot += cc[r];
        uint32_t run = firb_suffix<L>(tot) - tot;                             // the lanes above
#pragma unroll
        for (int r = T - 1; r >= 0; r--) { visit[r] = x.mag > run; run += cc[r]; }
    } else {
        // wide streams: the decrements can be large enough to wrap a 32-bit sum of 32 of them.  Saturating adds inside the lane
        // and ONE clamp of the lane total (instead of one per tap) keep every partial sum either exact or above any |err|
        uint32_t tot = cc[0];
#pragma unroll
        for (int r = 1; r < T; r++) tot = __builtin_elementwise_add_sat(tot, cc[r]);
        tot = min(tot, L == 16 ? 1u << 27 : 1u << 28);                        // (above any |err|; L of them do not wrap)
        uint32_t run = firb_suffix<L>(tot) - tot;
#pragma unroll
        for (int r = T - 1; r >= 0; r--) { visit[r] = x.mag > run; run = __builtin_elementwise_add_sat(run, cc[r]); }
    }
    // coef -= sign, sign = +-sgn(base - hist) (:325-327): the select sits on the median.  With the sign +-1 at hand it rides on
    // one multiply-add; the short queue entry only has the sign MASK s: coef + ((sd ^ s) - s) = ((sd ^ s) + (coef - s)), two
    // full-rate instructions (the 24-bit multiply-add issues at half rate once the SIMD is saturated -- which is where the
    // short form is used)
#pragma unroll
    for (int r = 0; r < T; r++) {
        const int sdv = visit[r] ? sd[r] : 0;
        if (SHORT) f.c[r] = xad_u32(sdv, smask, wsub(f.c[r], smask));
        else f.c[r] = mad_i24(sdv, x.sgn, f.c[r]);
    }
    f.h[PL] = __builtin_amdgcn_update_dpp(out, f.h[PL], L == 16 ? DPP_ROW_SHR1 : DPP_ROW_SHR2, 0xF, 0xF, false);
    f.base = (SPECIAL && f.delta) ? out : nb;
}

// Every case, on signed differences: the first sample (:260), the warm-up samples (:284-293), orders 0 and 31, streams that
// are switched off or have ended (`active` false), the chunk in which a stream ends.  Register rotation phase 0 on entry
// and exit.  `err` is the residual, `sgn` its sign mask (err >> 31).
template <int T, int L>
__device__ __forceinline__ void firb_step_masked(FirB<T>& f, int err, int i, bool active) {
    int src = f.h[0];
#pragma unroll
    for (int r = 1; r < T; r++) src = (f.bsel == r) ? f.h[r] : src;
    const int nb = __builtin_amdgcn_ds_bpermute(f.bpaddr, src);
    int d[T];
    int p = f.rnd0;
#pragma unroll
    for (int r = 0; r < T; r++) {
        d[r] = wsub(f.h[r], f.base);
        p = wadd(p, wmul(d[r], f.c[r]));
    }
    const int sum = firb_allreduce<L>(p);
    const int xu = wadd(wadd(sum >> f.q, f.base), err);
    const bool general = f.N >= 1 && f.N <= 30 && i > f.N;
    int out;
    if (i == 0 || f.n0) out = wadd(err, f.bias);                              // copies (:260-267)
    else if (!general)   // :268-293 (in the delta mode the base IS the previous sample, also behind steady-state steps, which keep no `prev`)
        out = (int)__builtin_amdgcn_ubfe((uint32_t)wadd(f.delta ? f.base : f.prev, err), 0u, (uint32_t)f.rss);
    else out = (int)__builtin_amdgcn_ubfe((uint32_t)xu, 0u, (uint32_t)f.rss);
    const int s = err >> 31;
    const uint32_t E = (uint32_t)((err ^ s) - s);
    uint32_t cc[T];
    uint32_t tot = 0;
#pragma unroll
    for (int r = 0; r < T; r++) {
        const int a = max(d[r], -d[r]);
        cc[r] = min(((uint32_t)(a + (s & f.qmask)) >> f.q) * f.w[r], 1u << 26);
        tot += cc[r];
    }
    uint32_t run = firb_suffix<L>(tot) - tot;
    const bool live = general && active;
#pragma unroll
    for (int r = T - 1; r >= 0; r--) {
        const int sdr = med3_i32(d[r], f.tlo[r], f.thi[r]);
        f.c[r] += (live && E > run) ? (sdr ^ s) - s : 0;
        run += cc[r];
    }
    const int h0 = __builtin_amdgcn_update_dpp(out, f.h[T - 1], L == 16 ? DPP_ROW_SHR1 : DPP_ROW_SHR2, 0xF, 0xF, false);
    if (active) {
#pragma unroll
        for (int r = T - 1; r >= 1; r--) f.h[r] = f.h[r - 1];
        f.h[0] = h0;
        f.base = f.delta ? out : nb;
        f.prev = out;
    }
}

// Everything a lane knows about its packet / stream after the header parse.
struct Meta {
    const uint8_t* base;   // 16-byte aligned-down packet start
    int64_t limit;         // readable bytes from base = up to the packet's last byte; what lies beyond reads as zero
    uint32_t size_bits_end; // bit position (from base) one past the packet's last bit
    int n;                 // samples per channel
    int status;
    int stereo, esc, ub, ss, nc, rss;
    int mixshift, mixweight;
    int N, q, rnd, ricemod, predtype;  // this row's channel
    uint32_t coefbit;      // bit position of this channel's first coefficient
    uint32_t ubit;         // bit position of the shift-byte block
    uint32_t ricebit;      // bit position where Rice stream A starts
    uint32_t rawbit;       // bit position of the first raw sample (escape packets)
    int out_bytes;
};


// Header parse (AlacFile.cs:435-475 / :584-641) for the stream (pkt, chan).  Every lane that needs a
// stream's facts calls this itself (a handful of cached loads).  cfg receives the stream cfg.
__device__ __forceinline__ Meta parse_meta(const alac_decode_params& p, uint32_t pkt, int chan, bool valid,
                                           alacgpu_cfg_dev& cfg) {
    Meta m;
    m.base = p.blob;
    m.limit = 0;
    m.size_bits_end = 0;
    m.n = 0;
    m.status = 0;
    m.stereo = m.esc = m.ub = 0;
    m.ss = 16;
    m.nc = 1;
    m.rss = 16;
    m.mixshift = m.mixweight = 0;
    m.N = m.q = m.ricemod = m.predtype = 0;
    m.rnd = 0;
    m.coefbit = m.ubit = m.ricebit = m.rawbit = 64;
    m.out_bytes = 0;
    cfg = p.cfgs[0];

    if (valid) {
        uint32_t ci = p.cfg_idx ? p.cfg_idx[pkt] : 0u;
        bool badcfg = ci >= p.n_cfgs;
        if (!badcfg) cfg = p.cfgs[ci];
        const uint64_t off = p.offsets[pkt];
        const uint32_t size = p.sizes[pkt];
        const uint32_t mis = (uint32_t)(off & 15u);
        m.base = p.blob + (off - mis);
        // readable bytes from base: up to the packet's last byte (reads past it give zeros, as in the oracle), and never
        // past the blob
        m.limit = min((int64_t)p.blob_limit - (int64_t)(off - mis), (int64_t)mis + (int64_t)size);
        const uint32_t bit0 = mis * 8u;
        m.size_bits_end = bit0 + size * 8u;
        m.ss = cfg.sample_size;
        m.nc = cfg.num_channels;
        const int ctor_ss = cfg.ctor_sample_size ? cfg.ctor_sample_size : cfg.sample_size;
        const int bytespersample = (ctor_ss / 8) * m.nc;                       // AlacFile.cs:19
        const uint32_t channels = peek_bits(m.base, m.limit, bit0, 3);         // :435
        const uint32_t hassize = peek_bits(m.base, m.limit, bit0 + 19, 1);     // :444,:586
        m.ub = (int)peek_bits(m.base, m.limit, bit0 + 20, 2);                  // :445,:587
        m.esc = (int)peek_bits(m.base, m.limit, bit0 + 22, 1);                 // :446,:588
        m.n = (int)cfg.max_samples_per_frame;                                  // :430
        uint32_t hdr_end = bit0 + 23;
        if (hassize) {
            m.n = (int)peek_bits(m.base, m.limit, bit0 + 23, 32);              // :451,:593
            hdr_end += 32;
        }
        m.out_bytes = (int)((uint32_t)m.n * (uint32_t)bytespersample);         // :436,:452,:718
        m.stereo = channels == 1;
        if (badcfg) m.status = ALACGPU_ST_UNSUPPORTED_PARAMS_D;
        else if (channels > 1) { m.status = ALACGPU_ST_UNSUPPORTED_ELEMENT_D; m.n = (int)cfg.max_samples_per_frame;
                                 m.out_bytes = (int)((uint32_t)m.n * (uint32_t)bytespersample); }
        else if (m.ss != 16 && m.ss != 24) m.status = ALACGPU_ST_UNSUPPORTED_SAMPLE_SIZE_D;
        else if (m.nc < 1 || m.nc > 2) m.status = ALACGPU_ST_UNSUPPORTED_ELEMENT_D;
        else if (m.n <= 0 || m.n > BUFFER_SIZE || (uint64_t)m.n * (uint64_t)m.nc > p.slot_ints)
            m.status = ALACGPU_ST_BAD_SAMPLE_COUNT_D;
        // a two-channel element in a one-channel stream comes out as its left channel (AlacFile.cs:353-354 with numchannels == 1:
        // every right sample is overwritten by the next left one); channel A is parked in the slot, which needs room for it
        else if (m.stereo && m.nc < 2 && (uint64_t)m.n * 2u > p.slot_ints) m.status = ALACGPU_ST_UNSUPPORTED_ELEMENT_D;
        else if (m.ss - m.ub * 8 < 8) m.status = ALACGPU_ST_UNSUPPORTED_PARAMS_D;
        m.rawbit = hdr_end;
        if (m.status == 0) {
            if (m.esc) {
                m.ub = 0;                                                       // :525,:697
            } else {
                m.rss = m.ss - m.ub * 8 + (m.stereo ? 1 : 0);                   // :454,:596
                if (m.stereo) {
                    m.mixshift = (int)peek_bits(m.base, m.limit, hdr_end, 8);       // :599
                    m.mixweight = (int)peek_bits(m.base, m.limit, hdr_end + 8, 8);  // :600 (unsigned)
                }
                uint32_t pa = hdr_end + 16;
                uint32_t ha = peek_bits(m.base, m.limit, pa, 16);               // :461-464 / :602-605
                int Na = (int)(ha & 31u);
                uint32_t pb = pa + 16 + 16u * Na;
                uint32_t hb = 0;
                int Nb = 0;
                if (m.stereo) {
                    hb = peek_bits(m.base, m.limit, pb, 16);                    // :618-621
                    Nb = (int)(hb & 31u);
                }
                uint32_t hh = chan == 0 ? ha : hb;
                m.predtype = (int)(hh >> 12) & 15;
                m.q = (int)(hh >> 8) & 15;
                m.ricemod = (int)(hh >> 5) & 7;
                m.N = chan == 0 ? Na : Nb;
                m.rnd = (int)(1u << ((m.q - 1) & 31));                          // 1 << (q-1), C# shift masking (:306)
                m.coefbit = (chan == 0 ? pa : pb) + 16;
                m.ubit = m.stereo ? pb + 16 + 16u * Nb : pb;
                m.ricebit = m.ubit + (uint32_t)m.n * (m.stereo ? 2u : 1u) * 8u * (uint32_t)m.ub;  // :476-482 / :634-641
            }
        }
    }

    return m;
}

}  // namespace alacdev
#endif

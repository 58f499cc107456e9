// alac_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the ALAC frame-decode path.
//
// What it computes: AlacFile.DecodeFrame (reference ALACDecoder/AlacFile.cs:428-719) for a batch of
// packets: element header parse, adaptive Golomb-Rice entropy decode (EntropyRiceDecode :214-252 =
// basterdised_rice_decompress), adaptive sign-LMS FIR reconstruction (PredictorDecompressFirAdapt
// :256-336 = predictor_decompress_fir_adapt) and mid/side un-mixing + store (Deinterlace16/24
// :338-421), integer only, bit-exact with the reference's C# int semantics.
//
// Mapping (v1, "fused row-per-stream"):
//   * one wave (64 lanes) = 4 rows of 16 lanes; a row owns one channel stream; a stereo packet is the
//     row pair (A,B); a wave therefore decodes TWO packets (a workgroup is one wave).
//   * the Rice state of a stream is replicated in its 16 lanes (row-uniform), so the residual is
//     available to every tap lane with no broadcast; the bitstream is staged (byte-swapped to
//     big-endian dwords) into a per-row LDS ring by coalesced 16-byte loads and read by a
//     three-dword sliding window whose next dword is prefetched one step ahead.
//   * the FIR keeps tap j's history sample and coefficient in lane j of the row (two registers per
//     lane when 16 < N <= 30): history shifts with one DPP row_shr, the dot product is a 4-step DPP
//     all-reduce, the sign-LMS early exit is a DPP suffix scan.
//   * channel B's Rice stream starts where A's ends, so the A rows first run a Rice-only pre-scan to
//     find that bit position; A and B are then decoded in lock step, un-mixed in registers every 16
//     samples and stored as coalesced int32 PCM.
// No MFMA: the path is integer/branchy, not a contraction.  No floating point anywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "alac_kernels.h"

namespace {

constexpr int RING_BYTES = 1024;          // per-row LDS ring
constexpr int RING_MASK = RING_BYTES - 1;
constexpr int FILL_CHUNK = 256;           // 16 lanes x 16 B
constexpr int BUFFER_SIZE = 16384;        // AlacFile.cs:28

// ---- C# int semantics: wrapping add/sub/mul ---------------------------------------------------------
__device__ __forceinline__ int wadd(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int wsub(int a, int b) { return (int)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int wmul(int a, int b) { return (int)((uint32_t)a * (uint32_t)b); }

// ---- DPP helpers (row = 16 lanes) --------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp0(int v) {  // out-of-row / invalid source lanes read 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
constexpr int DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_SHL1 = 0x101, DPP_ROW_SHL2 = 0x102, DPP_ROW_SHL4 = 0x104, DPP_ROW_SHL8 = 0x108;
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_ROR1 = 0x121;

__device__ __forceinline__ int row_allreduce_add(int v) {
    v = wadd(v, dpp0<DPP_QUAD_1032>(v));
    v = wadd(v, dpp0<DPP_QUAD_2301>(v));
    v = wadd(v, dpp0<DPP_ROW_HALF_MIRROR>(v));
    v = wadd(v, dpp0<DPP_ROW_MIRROR>(v));
    return v;
}
// inclusive suffix sum inside the row: lane l gets sum over lanes l..15
__device__ __forceinline__ int row_suffix_scan(int v) {
    v = wadd(v, dpp0<DPP_ROW_SHL1>(v));
    v = wadd(v, dpp0<DPP_ROW_SHL2>(v));
    v = wadd(v, dpp0<DPP_ROW_SHL4>(v));
    v = wadd(v, dpp0<DPP_ROW_SHL8>(v));
    return v;
}

// ---- slow-path bit access straight from global memory (header, shift bytes, escape samples) ----
__device__ __forceinline__ uint32_t load_be32(const uint8_t* base, int64_t byte_off, int64_t limit) {
    // byte_off is 4-aligned relative to a 16-aligned base
    if (byte_off < 0 || byte_off + 4 > limit) return 0;
    return __builtin_bswap32(*reinterpret_cast<const uint32_t*>(base + byte_off));
}
// nbits in 1..32, MSB-first field starting at bit `bitpos`
__device__ __forceinline__ uint32_t peek_bits(const uint8_t* base, int64_t limit, uint32_t bitpos, int nbits) {
    int64_t d = (int64_t)(bitpos >> 5) * 4;
    uint32_t hi = load_be32(base, d, limit), lo = load_be32(base, d + 4, limit);
    uint64_t w = ((uint64_t)hi << 32) | lo;
    w <<= (bitpos & 31);
    return (uint32_t)(w >> 32) >> (32 - nbits);
}

// ---- Rice reader state (row-uniform) -------------------------------------------------------------
struct Rice {
    uint32_t w0, w1, w2;  // three consecutive big-endian dwords; w2 is the prefetched one
    int rem;              // unconsumed bits left in w0, 0..31
    uint32_t next;        // byte offset (from the aligned packet base) of the dword after w2
    int hist;             // history            (AlacFile.cs:216)
    int signmod;          // signModifier       (:218)
    int zrun;             // zeros still to emit from the last run (:238-245)
};

__device__ __forceinline__ uint32_t rice_window(const Rice& s) {
    return __builtin_amdgcn_alignbit(s.w0, s.w1, s.rem);
}
__device__ __forceinline__ void rice_advance(Rice& s, int c, const uint32_t* ring) {
    int rem = s.rem - c;
    bool adv = rem < 0;
    s.rem = rem & 31;
    s.w0 = adv ? s.w1 : s.w0;
    s.w1 = adv ? s.w2 : s.w1;
    s.next += adv ? 4u : 0u;
    s.w2 = ring[((s.next - 4u) & RING_MASK) >> 2];
}
__device__ __forceinline__ uint32_t rice_bitpos(const Rice& s) { return (s.next - 12u) * 8u + 32u - (uint32_t)s.rem; }

// One EntropyDecodeValue (AlacFile.cs:193-212).  m = ((1<<k)-1) & mask, escape_bits = rss or 16.
__device__ __forceinline__ uint32_t rice_symbol(Rice& s, int k, uint32_t m, int escape_bits, const uint32_t* ring) {
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
};

// One output residual of EntropyRiceDecode (AlacFile.cs:219-251).  `remaining` = outputSize-1-outputCount.
// Sets *flags bit0 when a zero run would leave the reference's 16384-entry scratch, bit1 when history went negative.
__device__ __forceinline__ int rice_step(Rice& s, const RiceCfg& c, int remaining, int sample_idx, int* flags,
                                         const uint32_t* ring) {
    if (s.zrun > 0) {
        s.zrun--;
        return 0;
    }
    int t = (s.hist >> 9) + 3;
    int k = 31 - __clz(t);
    k = k < c.kmod ? k : c.kmod;                                             // :221-222
    uint32_t dv = rice_symbol(s, k, (1u << k) - 1u, c.rss, ring) + (uint32_t)s.signmod;  // :224
    s.signmod = 0;
    int r = (int)(dv >> 1) ^ -(int)(dv & 1u);                                // :225-226 (dv >= 0)
    int h = s.hist;
    h = (int)dv > 0xFFFF ? 0xFFFF : wsub(wadd(h, wmul((int)dv, c.hist_mult)), wmul(h, c.hist_mult) >> 9);  // :229
    if (h < 128 && remaining > 0) {                                          // :231
        if (h < 0) { *flags |= 2; h = 0; }
        s.signmod = 1;
        int k2 = (h == 0 ? 40 : __clz(h)) + ((h + 16) >> 6) - 24;           // :234 (clz(0) == 40 quirk)
        uint32_t bs = rice_symbol(s, k2, ((1u << (k2 & 31)) - 1u) & c.kmask, 16, ring);  // :236
        if ((uint32_t)sample_idx + bs > (uint32_t)(BUFFER_SIZE - 1)) *flags |= 1;        // :242 would throw
        s.zrun = bs > 0x7FFFFFFFu ? 0x7FFFFFFF : (int)bs;
        if (bs > 0xFFFF) s.signmod = 0;                                      // :246
        h = 0;                                                               // :248
    }
    s.hist = h;
    return r;
}

// ---- per-row LDS ring ------------------------------------------------------------------------------
// Tops the ring up with 256-byte chunks while there is room in front of the oldest live dword.
__device__ __forceinline__ void ring_fill(uint32_t* ring, uint32_t& filled, uint32_t next, const uint8_t* base,
                                          int64_t limit, int l, bool enable) {
    while (true) {
        bool need = enable && (filled + FILL_CHUNK <= (next - 12u) + RING_BYTES);
        if (!__builtin_amdgcn_ballot_w64(need)) break;
        if (need) {
            int64_t off = (int64_t)filled + l * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (off + 16 <= limit) v = *reinterpret_cast<const uint4*>(base + off);
            uint4 o = make_uint4(__builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z),
                                 __builtin_bswap32(v.w));
            *reinterpret_cast<uint4*>(&ring[(off & RING_MASK) >> 2]) = o;
            filled += FILL_CHUNK;
        }
    }
}

__device__ __forceinline__ void rice_init(Rice& s, uint32_t& filled, uint32_t startbit, int init_hist, uint32_t* ring,
                                          const uint8_t* base, int64_t limit, int l, bool enable) {
    uint32_t p = startbit - 1u;   // startbit >= 23 always
    uint32_t d = (p >> 5) * 4u;   // byte offset of the dword holding bit startbit-1
    s.rem = 31 - (int)(p & 31u);
    s.next = d + 12u;
    s.hist = init_hist;
    s.signmod = 0;
    s.zrun = 0;
    filled = d & ~(uint32_t)(FILL_CHUNK - 1);
    __syncthreads();
    ring_fill(ring, filled, s.next, base, limit, l, enable);
    __syncthreads();
    s.w0 = ring[((d) & RING_MASK) >> 2];
    s.w1 = ring[((d + 4u) & RING_MASK) >> 2];
    s.w2 = ring[((d + 8u) & RING_MASK) >> 2];
}

// ---- FIR state: tap j = l + 16*t lives in lane l, register t ---------------------------------------
template <int TPL>
struct Fir {
    int hist[TPL];  // out[i-1-j]
    int coef[TPL];  // predictorCoefTable[j] (0 for j >= N)
    int base;       // out[i-1-N]            (row-uniform)
    int prev;       // out[i-1]              (row-uniform)
};

template <int TPL>
__device__ __forceinline__ int fir_step(Fir<TPL>& f, int err, int i, int N, int q, int rnd, int rss, int l,
                                        int rowlane0) {
    int out;
    if (i == 0 || N == 0) {
        out = err;                                            // :260-267, first sample copies
    } else if (i <= N || N == 31) {
        out = __builtin_amdgcn_sbfe(wadd(f.prev, err), 0, rss);  // :268-293
    } else {
        int acc = 0;
        int d[TPL];
#pragma unroll
        for (int t = 0; t < TPL; t++) {
            d[t] = wsub(f.hist[t], f.base);                   // :303
            acc = wadd(acc, wmul(d[t], f.coef[t]));
        }
        int sum = row_allreduce_add(acc);
        int pred = wadd(wadd(rnd, sum) >> q, f.base);         // :306-308
        out = __builtin_amdgcn_sbfe(wadd(pred, err), 0, rss);  // :309-310
        if (err != 0) {                                       // :312-332 sign-LMS, parallel form
            // tap p is visited in order p = N-1 .. 0 while the running error keeps its sign; with
            // E = |err| and c_p = the magnitude tap p takes off it, tap p is visited iff
            // E - sum_{p' > p} c_p' > 0.
            const int sg = err > 0 ? 1 : -1;
            const uint32_t E = (uint32_t)(err > 0 ? err : -err);
            const int rnde = err < 0 ? (1 << q) - 1 : 0;      // (-a) >> q == -((a + 2^q - 1) >> q)
            uint32_t c[TPL];
#pragma unroll
            for (int t = 0; t < TPL; t++) {
                int a = d[t] < 0 ? -d[t] : d[t];
                int j = l + 16 * t;
                uint32_t w = j < N ? (uint32_t)(N - j) : 0u;
                uint32_t cc = ((uint32_t)(a + rnde) >> q) * w;
                c[t] = cc < (1u << 26) ? cc : (1u << 26);     // clamp: keeps the scan from wrapping, decisions unchanged
            }
            uint32_t upper = 0;  // contribution of the taps in higher registers (visited first)
#pragma unroll
            for (int t = TPL - 1; t >= 0; t--) {
                uint32_t incl = (uint32_t)row_suffix_scan((int)c[t]);
                uint32_t excl = incl - c[t] + upper;
                int j = l + 16 * t;
                bool visit = (j < N) && (E > excl);
                int sd = d[t] > 0 ? 1 : (d[t] < 0 ? -1 : 0);
                f.coef[t] += visit ? sg * sd : 0;             // coef[p] -= sign, sign = +-sgn(base - hist) (:325-327)
                if (t > 0) upper += (uint32_t)row_allreduce_add((int)c[t]);
            }
        }
    }
    // slide the history: tap N-1 becomes the next base, out enters at tap 0
    if (N >= 1 && N <= 30) {
        int src = (TPL > 1 && N > 16) ? f.hist[TPL - 1] : f.hist[0];
        f.base = __shfl(src, rowlane0 + ((N - 1) & 15), 64);
    }
#pragma unroll
    for (int t = TPL - 1; t >= 1; t--) {
        int carry = __builtin_amdgcn_update_dpp(0, f.hist[t - 1], DPP_ROW_ROR1, 0xF, 0xF, false);
        f.hist[t] = __builtin_amdgcn_update_dpp(carry, f.hist[t], DPP_ROW_SHR1, 0xF, 0xF, false);
    }
    f.hist[0] = __builtin_amdgcn_update_dpp(out, f.hist[0], DPP_ROW_SHR1, 0xF, 0xF, false);
    f.prev = out;
    return out;
}

// Everything a lane knows about its packet / stream after the header parse.
struct Meta {
    const uint8_t* base;   // 16-byte aligned-down packet start
    int64_t limit;         // readable bytes from base
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

template <int TPL>
__device__ void decode_wave(const Meta& m, const alacgpu_cfg_dev& cfg, bool valid, int row, int l, int lane, int chan,
                            uint32_t* ring, int32_t* pcm_slot, int32_t* st_out, uint32_t pkt) {
    const bool compressed = valid && m.status == 0 && !m.esc;
    const bool stream_on = compressed && (chan == 0 || m.stereo);
    const int n_row = stream_on ? m.n : 0;
    const int rowlane0 = lane & 48;
    int flags = 0;

    RiceCfg rc;
    rc.kmod = cfg.rice_kmodifier;
    rc.kmask = (1u << cfg.rice_kmodifier) - 1u;
    rc.hist_mult = m.ricemod * (cfg.rice_history_mult / 4);  // :483,:643,:653
    rc.rss = m.rss;

    Rice rs;
    uint32_t filled = 0;

    // ---- pre-scan: Rice-only pass over channel A of stereo packets to find where B starts ----
    const bool pre_on = stream_on && chan == 0 && m.stereo;
    const int n_pre = pre_on ? m.n : 0;
    int npre_max = max(max(__builtin_amdgcn_readlane(n_pre, 0), __builtin_amdgcn_readlane(n_pre, 16)),
                       max(__builtin_amdgcn_readlane(n_pre, 32), __builtin_amdgcn_readlane(n_pre, 48)));
    uint32_t bstart = m.ricebit;
    if (npre_max > 0) {
        rice_init(rs, filled, m.ricebit, cfg.rice_initial_history, ring, m.base, m.limit, l, pre_on);
        int dummy = 0;
        for (int i = 0; i < npre_max; i++) {
            if (i < n_pre) (void)rice_step(rs, rc, n_pre - 1 - i, i, &dummy, ring);
            if ((i & 15) == 15) {
                __syncthreads();
                ring_fill(ring, filled, rs.next, m.base, m.limit, l, pre_on);
                __syncthreads();
            }
        }
        bstart = rice_bitpos(rs);
    }
    uint32_t other = (uint32_t)__shfl((int)bstart, lane ^ 16, 64);
    const uint32_t startbit = (chan == 1) ? other : m.ricebit;

    // ---- main pass ----
    int nmax = max(max(__builtin_amdgcn_readlane(n_row, 0), __builtin_amdgcn_readlane(n_row, 16)),
                   max(__builtin_amdgcn_readlane(n_row, 32), __builtin_amdgcn_readlane(n_row, 48)));
    // escape packets have no Rice/FIR work but still need the output stage
    const int n_out = (valid && m.status == 0) ? m.n : 0;
    int nout_max = max(max(__builtin_amdgcn_readlane(n_out, 0), __builtin_amdgcn_readlane(n_out, 16)),
                       max(__builtin_amdgcn_readlane(n_out, 32), __builtin_amdgcn_readlane(n_out, 48)));
    if (nmax > 0) rice_init(rs, filled, startbit, cfg.rice_initial_history, ring, m.base, m.limit, l, stream_on);

    Fir<TPL> f;
#pragma unroll
    for (int t = 0; t < TPL; t++) {
        int j = l + 16 * t;
        f.hist[t] = 0;
        int cv = 0;
        if (stream_on && j < m.N) cv = (int)(int16_t)peek_bits(m.base, m.limit, m.coefbit + 16u * j, 16);  // :468-474
        f.coef[t] = cv;
    }
    f.base = 0;
    f.prev = 0;

    int cap = 0;
    for (int i0 = 0; i0 < nout_max; i0 += 16) {
        if (i0 < nmax) {
            int iend = min(16, nmax - i0);
            for (int ii = 0; ii < iend; ii++) {
                int i = i0 + ii;
                if (i < n_row) {
                    int err = rice_step(rs, rc, n_row - 1 - i, i, &flags, ring);
                    int out = fir_step<TPL>(f, err, i, m.N, m.q, m.rnd, m.rss, l, rowlane0);
                    cap = (l == ii) ? out : cap;
                }
            }
            __syncthreads();
            ring_fill(ring, filled, rs.next, m.base, m.limit, l, stream_on && i0 + 16 < n_row);
            __syncthreads();
        }
        // ---- output stage: 16 sample frames per packet, un-mix + shift bytes + coalesced store ----
        int i = i0 + l;
        bool live = i < n_out;
        int mine = cap;
        if (live && m.esc && (chan == 0 || m.stereo)) {  // raw samples (:500-525 / :665-699)
            uint32_t bp = m.rawbit + (uint32_t)((i * (m.stereo ? 2 : 1) + chan) * m.ss);
            int v = (int)peek_bits(m.base, m.limit, bp, m.ss);
            mine = __builtin_amdgcn_sbfe(v, 0, m.ss);
        }
        int partner = __shfl(mine, lane ^ 16, 64);
        if (live) {
            int a = chan == 0 ? mine : partner, b = chan == 0 ? partner : mine;
            int val;
            if (m.stereo) {
                int left, right;
                if (m.mixweight != 0) {                                 // :344-351
                    right = wsub(a, wmul(b, m.mixweight) >> (m.mixshift & 31));
                    left = wadd(right, b);
                } else {
                    left = a;
                    right = b;
                }
                val = chan == 0 ? left : right;
            } else {
                val = chan == 0 ? a : 0;                                 // :531-541: silent second channel
            }
            if (m.ss == 24) {
                if (m.ub != 0 && !m.esc && (chan == 0 || m.stereo)) {   // :381-388 / :549-554
                    uint32_t bp = m.ubit + (uint32_t)((i * (m.stereo ? 2 : 1) + chan) * 8 * m.ub);
                    uint32_t sb = peek_bits(m.base, m.limit, bp, 8 * m.ub);
                    val = (int)(((uint32_t)val << (8 * m.ub)) | sb);
                }
                val = __builtin_amdgcn_sbfe(val, 0, 24);
            }
            if (chan < m.nc) pcm_slot[(int64_t)i * m.nc + chan] = val;
        }
    }

    // ---- status: same priority order as the oracle / the reference's control flow ----
    int fl_other = __shfl(flags, lane ^ 16, 64);
    int pt_other = __shfl(m.predtype, lane ^ 16, 64);
    int N_other = __shfl(m.N, lane ^ 16, 64);
    uint32_t endbit = rice_bitpos(rs);
    uint32_t end_other = (uint32_t)__shfl((int)endbit, lane ^ 16, 64);
    if (valid && l == 0 && chan == 0) {
        int st = m.status;
        if (st == 0 && !m.esc) {
            const int nch = m.stereo ? 2 : 1;
            for (int c = 0; c < nch && st == 0; c++) {
                int fl = c == 0 ? flags : fl_other;
                int pt = c == 0 ? m.predtype : pt_other;
                int Nc = c == 0 ? m.N : N_other;
                if (fl & 1) st = ALACGPU_ST_OVERRUN_D;
                else if (fl & 2) st = ALACGPU_ST_UNSUPPORTED_PARAMS_D;
                else if (pt != 0) st = ALACGPU_ST_UNSUPPORTED_PREDTYPE_D;
                else if (Nc == 0 && m.n > 4096) st = ALACGPU_ST_REF_THROWS_D;
            }
            uint32_t last = m.stereo ? end_other : endbit;
            if (st == 0 && last > m.size_bits_end) st = ALACGPU_ST_OVERRUN_D;
        } else if (st == 0 && m.esc) {
            uint32_t last = m.rawbit + (uint32_t)(m.n * (m.stereo ? 2 : 1) * m.ss);
            if (last > m.size_bits_end) st = ALACGPU_ST_OVERRUN_D;
        }
        st_out[pkt] = st;
    }
}

}  // namespace

extern "C" __global__ __launch_bounds__(64) void alac_decode_packets_kernel(alac_decode_params p) {
    __shared__ __attribute__((aligned(16))) uint32_t ring_all[4][RING_BYTES / 4];
    const int lane = threadIdx.x;
    const int row = lane >> 4, l = lane & 15, chan = row & 1;
    const uint32_t pkt = blockIdx.x * 2u + (uint32_t)(row >> 1);
    const bool valid = pkt < p.n_packets;

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
    alacgpu_cfg_dev cfg = p.cfgs[0];

    if (valid) {
        uint32_t ci = p.cfg_idx ? p.cfg_idx[pkt] : 0u;
        bool badcfg = ci >= p.n_cfgs;
        if (!badcfg) cfg = p.cfgs[ci];
        const uint64_t off = p.offsets[pkt];
        const uint32_t size = p.sizes[pkt];
        const uint32_t mis = (uint32_t)(off & 15u);
        m.base = p.blob + (off - mis);
        m.limit = (int64_t)p.blob_limit - (int64_t)(off - mis);
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
        else if (m.stereo && m.nc < 2) m.status = ALACGPU_ST_UNSUPPORTED_ELEMENT_D;
        else if (m.nc < 1 || m.nc > 2) m.status = ALACGPU_ST_UNSUPPORTED_ELEMENT_D;
        else if (m.n <= 0 || m.n > BUFFER_SIZE || (uint64_t)m.n * (uint64_t)m.nc > p.slot_ints)
            m.status = ALACGPU_ST_BAD_SAMPLE_COUNT_D;
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
        if (l == 0 && chan == 0) {
            if (p.out_bytes) p.out_bytes[pkt] = m.out_bytes;
            if (p.out_samples) p.out_samples[pkt] = m.n;
        }
    }

    int32_t* pcm_slot = p.pcm_out + (int64_t)pkt * p.slot_ints;
    uint32_t* ring = ring_all[row];
    // two tap registers per lane only when some stream in this wave needs more than 16 taps
    const bool wide = valid && m.status == 0 && !m.esc && m.N > 16 && m.N <= 30 && (chan == 0 || m.stereo);
    if (__builtin_amdgcn_ballot_w64(wide))
        decode_wave<2>(m, cfg, valid, row, l, lane, chan, ring, pcm_slot, p.status, pkt);
    else
        decode_wave<1>(m, cfg, valid, row, l, lane, chan, ring, pcm_slot, p.status, pkt);
}

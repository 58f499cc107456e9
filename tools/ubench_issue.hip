// ubench_issue.hip -- single-wave issue/latency microbenchmarks for gfx950 (design input for the
// ALAC kernels, whose hot loops are dependent integer chains).  Prints cycles per instruction for a
// wave running alone and wall time when 1/2/4/8 such waves share each SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITERS 256

template <int K>
__global__ __launch_bounds__(64) void kern(unsigned long long* out, int* sink, int seed) {
    __shared__ int lds[1024];
    int a = seed + threadIdx.x, b = seed * 3 + 1, c = seed + 7, d = seed + 11;
    int s0 = seed, s1 = seed + 1;
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = ((i * 7 + 3) & 1023) * 4;
    __syncthreads();
    int p = (threadIdx.x * 4) & 4095;
    if (K == 8) { if (threadIdx.x >= 32) { a = 0; } }
    unsigned long long t0 = clock64();
    if (K == 8 && threadIdx.x >= 32) goto done;  // exec = low half only
    for (int it = 0; it < ITERS; it++) {
        if (K == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (K == 1) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 2) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (K == 3) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (K == 4) { REP16(asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));) }
        if (K == 5) { REP16(asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");) }
        if (K == 6) { REP16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(p));) }
        if (K == 7) { REP16(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(p));) }
        if (K == 8) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (K == 9) { REP16(asm volatile("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, %3" : "+v"(a), "+s"(s0) : "v"(b), "s"(s1) : "scc");) }
        if (K == 10) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 11) { REP16(asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");) }
        if (K == 12) { REP16(asm volatile("v_readlane_b32 %1, %0, 3\n v_add_u32 %0, %0, %1" : "+v"(a), "+s"(s0));) }
        if (K == 13) { REP16(asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (K == 20) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 21) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 22) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %4, %4\n v_mad_u32_u24 %2, %2, %4, %4\n v_mad_u32_u24 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 23) { REP16(asm volatile("v_bfe_u32 %0, %0, %4, 5\n v_bfe_u32 %1, %1, %4, 5\n v_bfe_u32 %2, %2, %4, 5\n v_bfe_u32 %3, %3, %4, 5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 24) { REP16(asm volatile("v_alignbit_b32 %0, %0, %4, %4\n v_alignbit_b32 %1, %1, %4, %4\n v_alignbit_b32 %2, %2, %4, %4\n v_alignbit_b32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 25) { REP16(asm volatile("v_sad_u32 %0, %0, %4, %4\n v_sad_u32 %1, %1, %4, %4\n v_sad_u32 %2, %2, %4, %4\n v_sad_u32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 26) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed) : "vcc");) }
        if (K == 27) { REP16(asm volatile("v_bitop3_b32 %0, %0, %4, %4 bitop3:0x26\n v_bitop3_b32 %1, %1, %4, %4 bitop3:0x26\n v_bitop3_b32 %2, %2, %4, %4 bitop3:0x26\n v_bitop3_b32 %3, %3, %4, %4 bitop3:0x26" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 28) { REP16(asm volatile("v_med3_i32 %0, %0, %4, %4\n v_med3_i32 %1, %1, %4, %4\n v_med3_i32 %2, %2, %4, %4\n v_med3_i32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 29) { REP16(asm volatile("v_max3_u32 %0, %0, %4, %4\n v_max3_u32 %1, %1, %4, %4\n v_max3_u32 %2, %2, %4, %4\n v_max3_u32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 30) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 31) { REP16(asm volatile("v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n v_add3_u32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 32) { REP16(asm volatile("v_ffbh_u32 %0, %0\n v_ffbh_u32 %1, %1\n v_ffbh_u32 %2, %2\n v_ffbh_u32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 33) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 34) { REP16(asm volatile("v_sub_u32 %0, %0, 1 clamp\n v_sub_u32 %1, %1, 1 clamp\n v_sub_u32 %2, %2, 1 clamp\n v_sub_u32 %3, %3, 1 clamp" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 35) { REP16(asm volatile("v_and_or_b32 %0, %0, %4, %4\n v_and_or_b32 %1, %1, %4, %4\n v_and_or_b32 %2, %2, %4, %4\n v_and_or_b32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 36) { REP16(asm volatile("v_bfe_i32 %0, %0, 0, 7\n v_bfe_i32 %1, %1, 0, 7\n v_bfe_i32 %2, %2, 0, 7\n v_bfe_i32 %3, %3, 0, 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 37) { REP16(asm volatile("v_subb_co_u32 %0, vcc, %0, %4, vcc\n v_subb_co_u32 %1, vcc, %1, %4, vcc\n v_subb_co_u32 %2, vcc, %2, %4, vcc\n v_subb_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed) : "vcc");) }
        if (K == 38) { REP16(asm volatile("v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (K == 39) { REP16(asm volatile("v_add_u32_dpp %0, %0, %0 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %1, %1, %1 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %2, %2, %2 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %3, %3, %3 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        // control flow (round 2): what a hop costs a wave that is alone on its SIMD.  The figures are per pair / group below.
        if (K == 40) { REP16(asm volatile("v_add_u32 %0, %0, %1\n s_branch 1f\n s_nop 0\n s_nop 0\n1:" : "+v"(a) : "v"(b));) }                 // taken branch, target in the same cache line
        if (K == 41) { REP16(asm volatile("v_add_u32 %0, %0, %1\n s_branch 1f\n .fill 256, 4, 0xbf800000\n1:" : "+v"(a) : "v"(b));) }       // taken branch over 1 KiB of s_nop
        if (K == 42) { REP16(asm volatile("v_cmp_lt_i32 vcc, %0, %1\n s_cbranch_vccnz 1f\n s_nop 0\n1:\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b) : "vcc");) }   // compare -> conditional branch, NOT taken (a > b)
        if (K == 43) { REP16(asm volatile("v_cmp_gt_i32 vcc, %0, %1\n s_cbranch_vccnz 1f\n .fill 64, 4, 0xbf800000\n1:\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b) : "vcc");) }   // ... taken, over 256 B
        if (K == 44) { REP16(asm volatile("v_cmp_lt_i32 vcc, %0, %1\n s_cmp_lg_u64 vcc, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b) : "vcc", "scc");) }   // ballot != 0 idiom, not taken
        if (K == 45) { REP16(asm volatile("v_add_u32 %0, %0, %1\n s_nop 0" : "+v"(a) : "v"(b));) }                                              // reference: the add with one s_nop
        if (K == 14) { REP16(asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(*(long long*)&lds[0]) : "v"(b));) }
    }
done:
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a + b + c + d + s0 + p == 0x12345678) sink[0] = a;
}

template <int K>
void run(const char* name, int instr_per_rep, unsigned long long* d_out, int* d_sink) {
    printf("%-28s", name); fflush(stdout);
    for (int waves_per_simd : {0, 1, 2, 4, 8}) {
        int grid = waves_per_simd == 0 ? 1 : 1024 * waves_per_simd;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        kern<K><<<grid, 64>>>(d_out, d_sink, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<K><<<grid, 64>>>(d_out, d_sink, 1);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), d_out, grid * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += (double)v; avg /= grid;
        double n = (double)ITERS * 16 * instr_per_rep;
        printf(" | w/simd=%d: %6.2f cyc/instr (wall %7.1f us)", waves_per_simd, avg / n, ms * 1e3);
    }
    printf("\n"); fflush(stdout);
}

int main() {
    unsigned long long* d_out; int* d_sink;
    hipMalloc(&d_out, 8192 * 8 * 2); hipMalloc(&d_sink, 64);
    printf("clock64 ticks per instruction (one wave per block, 64 threads); w/simd=0 means a single wave on the chip\n");
    run<0>("v_add dep chain", 1, d_out, d_sink);
    run<1>("v_add 4 indep chains", 4, d_out, d_sink);
    run<2>("v_mul_lo_u32 dep", 1, d_out, d_sink);
    run<10>("v_mul_lo_u32 4 indep", 4, d_out, d_sink);
    run<3>("v_mul_u32_u24 dep", 1, d_out, d_sink);
    run<4>("s_nop1+v_add_dpp dep", 1, d_out, d_sink);
    run<5>("s_add dep chain", 1, d_out, d_sink);
    run<9>("v_add+s_add interleaved", 2, d_out, d_sink);
    run<11>("v_cmp+v_cndmask dep", 2, d_out, d_sink);
    run<12>("v_readlane+v_add dep", 2, d_out, d_sink);
    run<13>("v_alignbit dep", 1, d_out, d_sink);
    run<6>("ds_read_b32 dep (+wait)", 1, d_out, d_sink);
    run<7>("ds_bpermute dep (+wait)", 1, d_out, d_sink);
    run<8>("v_add dep, exec low half", 1, d_out, d_sink);
    printf("throughput, 4 independent chains of one instruction:\n");
    run<20>("4x v_add_u32", 4, d_out, d_sink);
    run<21>("4x v_mul_lo_u32", 4, d_out, d_sink);
    run<22>("4x v_mad_u32_u24", 4, d_out, d_sink);
    run<23>("4x v_bfe_u32", 4, d_out, d_sink);
    run<24>("4x v_alignbit_b32", 4, d_out, d_sink);
    run<25>("4x v_sad_u32", 4, d_out, d_sink);
    run<26>("4x v_cndmask_b32 (vcc)", 4, d_out, d_sink);
    run<27>("4x v_bitop3_b32", 4, d_out, d_sink);
    run<28>("4x v_med3_i32", 4, d_out, d_sink);
    run<29>("4x v_max3_u32", 4, d_out, d_sink);
    run<30>("4x v_lshl_add_u32", 4, d_out, d_sink);
    run<31>("4x v_add3_u32", 4, d_out, d_sink);
    run<32>("4x v_ffbh_u32", 4, d_out, d_sink);
    run<33>("4x v_mul_u32_u24", 4, d_out, d_sink);
    run<34>("4x v_sub_u32 clamp (VOP3)", 4, d_out, d_sink);
    run<35>("4x v_and_or_b32", 4, d_out, d_sink);
    run<36>("4x v_bfe_i32", 4, d_out, d_sink);
    run<37>("4x v_subb_co_u32", 4, d_out, d_sink);
    run<38>("4x v_max_i32", 4, d_out, d_sink);
    run<39>("4x v_add_u32_dpp (own chain; hazard nops NOT added)", 4, d_out, d_sink);
    printf("control flow, per group (the v_add included):\n");
    run<45>("v_add + s_nop (reference)", 1, d_out, d_sink);
    run<40>("v_add + s_branch (near)", 1, d_out, d_sink);
    run<41>("v_add + s_branch over 1 KiB", 1, d_out, d_sink);
    run<42>("v_cmp + cbranch NOT taken + v_add", 1, d_out, d_sink);
    run<43>("v_cmp + cbranch taken 256 B + v_add", 1, d_out, d_sink);
    run<44>("v_cmp + s_cmp + cbranch not taken + v_add", 1, d_out, d_sink);
    return 0;
}

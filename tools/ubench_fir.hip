// ubench_fir.hip -- cycles per sample of the P8 reconstruction step (fir8_step, alac_device.h) for one wave running
// alone on its SIMD and with 2 / 3 such waves per SIMD (design input: is the FIR wave bound by its dependent chain,
// by its instruction count or by sharing the SIMD?).  Residuals come from LDS as in the kernel.
//   hipcc -O3 --offload-arch=gfx950 -I../include -I../alac.net_amd/csrc -o ubench_fir ubench_fir.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#include "alac_device.h"

using namespace alacdev;

#define STEPS 4096

__global__ __launch_bounds__(64) void kern(unsigned long long* out, int* sink, uint32_t seed) {
    __shared__ int resq[2][16][8];
    const int lane = threadIdx.x, l = lane & 15, par = l & 1, j = l >> 1, row = lane >> 4;
    for (int i = lane; i < 2 * 16 * 8; i += 64) (&resq[0][0][0])[i] = (int)((i * 2654435761u + seed) >> 20) - 2048;
    __syncthreads();
    Fir8Lane f;
    f.hist = 0; f.coef = (j < 8) ? 100 - 20 * j : 0; f.base = 0; f.prev = 0;
    f.q = 9; f.rnd = 1 << 8; f.rss = 17; f.qmask = (1 << 9) - 1; f.N = 8;
    f.tlo = -1; f.thi = 1; f.w = (uint32_t)(8 - j);
    f.bpaddr = ((lane & 48) + 2 * 7 + par) * 4;
    const int g = 2 * row + par;
    const unsigned long long t0 = clock64();
    for (int c = 0; c < STEPS / 16; c++) {
        const int* q = &resq[c & 1][0][g];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            int err = q[(8 * half) * 8];
#pragma unroll
            for (int ii = 0; ii < 8; ii++) {
                const int en = q[(8 * half + (ii < 7 ? ii + 1 : ii)) * 8];
                fir8_step<false>(f, err, 16 + ii, true);
                err = en;
            }
        }
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    if (f.hist + f.coef == 0x12345678) sink[0] = f.base;
}

void run(const char* name, unsigned long long* d_out, int* d_sink) {
    printf("%-34s", name);
    for (int wps : {0, 1, 2, 3}) {
        const int grid = wps == 0 ? 1 : 1024 * wps;
        kern<<<grid, 64>>>(d_out, d_sink, 1);
        hipDeviceSynchronize();
        kern<<<grid, 64>>>(d_out, d_sink, 2);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), d_out, grid * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += (double)v; avg /= grid;
        printf(" | w/simd=%d: %6.1f cyc/sample", wps, avg / STEPS);
    }
    printf("\n"); fflush(stdout);
}

int main() {
    unsigned long long* d_out; int* d_sink;
    hipMalloc(&d_out, 8192 * 8); hipMalloc(&d_sink, 64);
    printf("clock64 ticks per sample of fir8_step, one wave per block; w/simd=0: a single wave on the chip\n");
    run("fir8_step", d_out, d_sink);
    return 0;
}

// tools/pcie_probe.hip -- host link speeds as the host-buffer path sees them (build: hipcc --offload-arch=gfx950 -O2 -o tools/pcie_probe tools/pcie_probe.hip)
// hipMemcpy H2D / D2H with pageable and pinned memory, and a copy KERNEL that writes device data straight into pinned
// host memory (what a fused "download" stage would do).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void copy_kernel(uint4* dst, const uint4* src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t sizes[2] = {45u << 20, 134u << 20};
    void* d; CK(hipMalloc(&d, 134u << 20));
    void* pin; CK(hipHostMalloc(&pin, 134u << 20, hipHostMallocDefault));
    void* pag = std::malloc(134u << 20); std::memset(pag, 1, 134u << 20); std::memset(pin, 1, 134u << 20);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (size_t sz : sizes) {
        for (int kind = 0; kind < 2; kind++) {
            void* h = kind ? pin : pag;
            for (int dir = 0; dir < 2; dir++) {
                double best = 1e9;
                for (int r = 0; r < 5; r++) {
                    double t = now();
                    if (dir == 0) CK(hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, s)); else CK(hipMemcpyAsync(h, d, sz, hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                    best = std::min(best, now() - t);
                }
                std::printf("%s %s %zu MiB: %.3f ms = %.1f GB/s\n", dir ? "D2H" : "H2D", kind ? "pinned  " : "pageable", sz >> 20, best * 1e3, sz / best / 1e9);
            }
        }
        for (int blocks : {64, 256, 1024}) {
            double best = 1e9;
            for (int r = 0; r < 5; r++) {
                double t = now();
                hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s, (uint4*)pin, (const uint4*)d, sz / 16);
                CK(hipStreamSynchronize(s));
                best = std::min(best, now() - t);
            }
            std::printf("copy kernel device -> pinned host, %d blocks, %zu MiB: %.3f ms = %.1f GB/s\n", blocks, sz >> 20, best * 1e3, sz / best / 1e9);
        }
        for (int blocks : {256}) {
            double best = 1e9;
            for (int r = 0; r < 5; r++) {
                double t = now();
                hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s, (uint4*)d, (const uint4*)pin, sz / 16);
                CK(hipStreamSynchronize(s));
                best = std::min(best, now() - t);
            }
            std::printf("copy kernel pinned host -> device, %d blocks, %zu MiB: %.3f ms = %.1f GB/s\n", blocks, sz >> 20, best * 1e3, sz / best / 1e9);
        }
    }
    return 0;
}

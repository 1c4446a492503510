// Dev probe: how fast are LDS atomics?  Every wave issues N x 16 ds_add_u32 (no return) to an LDS histogram laid out like
// hist_partial_kernel's ([bin][32 copies], lane l on copy l & 31) with bins from a register LCG (no memory traffic), in four
// address patterns: 0 = random bins (the histogram's), 1 = one bin for all lanes (constant image), 2 = lane-private word
// (no two lanes ever share a bank... 64 lanes on 32 banks: two per bank), 3 = random bins, 2 copies only (heavy conflicts).
// Prints lanes (= atomics) per clock and CU at the measured kernel time.  hipcc --offload-arch=gfx950 -O3 -o ldsatomic ldsatomic.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(512) void k(uint32_t* out, int iters, int mode)
{
    __shared__ uint32_t h[256 * 32];
    for (int i = threadIdx.x; i < 256 * 32; i += 512) h[i] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const unsigned copy = threadIdx.x & 31u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x = x * 1664525u + 1013904223u;
            uint32_t v = x >> 8;                                   // 24 random bits: four 6..8-bit bins
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t bin = (v >> (6 * q)) & 255u;
                uint32_t a;
                if (mode == 0) a = (bin << 5) | copy;
                else if (mode == 1) a = (77u << 5) | copy;
                else if (mode == 2) a = ((threadIdx.x & 63u) << 5) | copy;      // a word of its own per lane
                else a = (bin << 5) | (copy & 1u);
                atomicAdd(&h[a], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t s = 0;
    for (int i = threadIdx.x; i < 256 * 32; i += 512) s += h[i];
    if (s == 0xffffffffu) out[0] = s;
}
int main()
{
    uint32_t* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4, iters = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, iters, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double atomics = (double)blocks * 512 * iters * 16;
        printf("mode %d: %.3f ms, %.1f G atomics/s, %.2f lanes per clock and CU at 2.4 GHz\n", mode, ms, atomics / ms / 1e6, atomics / (ms * 1e-3) / 256 / 2.4e9);
    }
    return 0;
}

// Dev microbenchmark: LDS atomic rate.  Every lane adds to its own bank (address = bin * 32 + (lane & 31) dwords, bins from a
// per-lane LCG: the layout of ipx_hist.hip), N adds per lane, no memory traffic.  Variants: ds_add_u32 without / with return,
// a plain ds_write_b32 and ds_read_b32 for comparison, and a [bin][64] layout (one copy per LANE: lanes l and l + 32 differ).
// hipcc --offload-arch=gfx950 -O3 -o ldsatomic ldsatomic.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template <int MODE, int COPIES>
__global__ __launch_bounds__(512) void k(uint32_t* out, int iters, uint32_t seed)
{
    extern __shared__ uint32_t h[];
    for (int i = threadIdx.x; i < 256 * COPIES; i += 512) h[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & (COPIES - 1);
    uint32_t s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u, acc = 0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        uint32_t w = s;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t bin = (w >> (8 * j)) & 255u;
            uint32_t* p = h + bin * COPIES + lane;
            if (MODE == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // result unused: no-return form
            else if (MODE == 1) acc += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 2) *(volatile uint32_t*)p = w;
            else acc += *(volatile uint32_t*)p;
        }
    }
    __syncthreads();
    uint32_t t = acc;
    for (int i = threadIdx.x; i < 256 * COPIES; i += 512) t += h[i];
    if (t == 0x12345678u) out[blockIdx.x] = t;
}

// the histogram kernel's shape: a block streams a contiguous slice (16-byte loads, one step ahead) and counts every byte
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(512) void hk(const u32x4* __restrict__ src, size_t n16, uint32_t* out)
{
    __shared__ uint32_t h[256 * 32];
    for (int i = threadIdx.x; i < 256 * 32; i += 512) h[i] = 0;
    __syncthreads();
    const uint32_t copy = threadIdx.x & 31u;
    const size_t per = n16 / gridDim.x, b0 = per * blockIdx.x;
    u32x4 v[U], nv[U];
#pragma unroll
    for (int j = 0; j < U; ++j) v[j] = __builtin_nontemporal_load(src + b0 + j * 512 + threadIdx.x);
    for (size_t i = 0; i < per; i += 512 * U) {
        const size_t nx = i + 512 * U < per ? i + 512 * U : i;
#pragma unroll
        for (int j = 0; j < U; ++j) nv[j] = __builtin_nontemporal_load(src + b0 + nx + j * 512 + threadIdx.x);
#pragma unroll
        for (int j = 0; j < U; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t x = v[j][k];
                __hip_atomic_fetch_add(h + (((x & 255u) << 5) | copy), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(h + ((((x >> 8) & 255u) << 5) | copy), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(h + ((((x >> 16) & 255u) << 5) | copy), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(h + (((x >> 24)) << 5 | copy), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = nv[j];
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        uint32_t sum = 0;
        for (int j = 0; j < 32; ++j) sum += h[(threadIdx.x << 5) | ((j + threadIdx.x) & 31u)];
        out[blockIdx.x * 256 + threadIdx.x] = sum;
    }
}

template <int U>
int runh(const u32x4* src, size_t bytes, int blocks, uint32_t* out)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((hk<U>), dim3(blocks), dim3(512), 0, 0, src, bytes / 16, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 2) printf("stream + count, U=%d, %4d blocks, %4zu MB: %7.1f us  %5.2f TB/s\n", U, blocks, bytes >> 20, ms * 1e3, bytes / ms / 1e9);
    }
    return 0;
}

template <int MODE, int COPIES>
int run(const char* name, uint32_t* out)
{
    const int blocks = 256 * 4, iters = 4096;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<MODE, COPIES>), dim3(blocks), dim3(512), 256 * COPIES * 4, 0, out, iters, 1u + rep);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 2) {
            const double ops = (double)blocks * 512 * iters * 4;
            printf("%-44s %8.1f us  %6.2f Tops/s  %5.2f lanes/clk/CU (2.4 GHz, 256 CUs)\n", name, ms * 1e3, ops / ms / 1e9, ops / (ms * 1e-3) / 2.4e9 / 256);
        }
    }
    return 0;
}

int main()
{
    uint32_t* out; CK(hipMalloc(&out, 8 << 20));
    {
        const size_t bytes = (size_t)128 << 20;
        u32x4* src; CK(hipMalloc(&src, bytes + (1 << 20)));
        CK(hipMemset(src, 0x5a, bytes + (1 << 20)));
        std::vector<uint32_t> hbuf(bytes / 4);
        uint32_t s = 12345u;
        for (auto& w : hbuf) { s = s * 1664525u + 1013904223u; w = s; }
        CK(hipMemcpy(src, hbuf.data(), bytes, hipMemcpyHostToDevice));
        for (int blocks : {256, 512, 1024, 2048, 4096}) { runh<1>(src, bytes, blocks, out); runh<2>(src, bytes, blocks, out); runh<4>(src, bytes, blocks, out); }
    }
    run<0, 32>("ds_add_u32 no return, [bin][32]", out);
    run<1, 32>("ds_add_rtn_u32, [bin][32]", out);
    run<2, 32>("ds_write_b32, [bin][32]", out);
    run<3, 32>("ds_read_b32, [bin][32]", out);
    run<0, 64>("ds_add_u32 no return, [bin][64]", out);
    run<2, 64>("ds_write_b32, [bin][64]", out);
    return 0;
}

// ipx_stats.hip -- numpy's float32 sum of a table applied to a uint8 image, bit for bit.
//
// mode=denoise_manual (server.py:197-200) reduces `flat = img.flatten().astype(np.float32)` with np.mean / np.var /
// np.std.  Those are float32 reductions whose value depends on the ORDER of the additions once partial sums pass 2^24,
// and numpy's order is fixed: the ufunc machinery feeds np.add.reduce buffers of 8192 elements (np.getbufsize()), adds
// each buffer's PAIRWISE sum to the running total in sequence, and the pairwise sum works on blocks of <= 128 values
// (8 interleaved accumulators, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), remainder added serially), halving
// larger inputs at n/2 rounded down to a multiple of 8.  Both np.mean (values v) and np.var (values fl(fl(v-mean)^2))
// are sums of a 256-entry float table applied to the pixels, so one entry point serves both:
//   full 8192-element buffers: one 64-thread block each; thread t sums the 128-element block t in numpy's order, the 64
//     block sums are combined as the balanced binary tree the halving produces; one float per buffer goes back;
//   the host adds the buffer sums in sequence (float32) and evaluates the ragged last buffer itself (same recursion).
// Verified against numpy in tests/ (oracle/oracle.py pairwise_sum_f32 == np.add.reduce on this image's numpy).
#include "ipx_common.h"

// numpy rounds after every operation: a*b+c must never be fused into one FMA (in this HIP version the __f*_rn
// intrinsics are plain operators, which the default -ffp-contract=fast would contract)
#pragma clang fp contract(off)
#include <vector>

namespace {

constexpr int kBuf = 8192, kLeaf = 128;
struct TableArg { float f[256]; };

__global__ __launch_bounds__(64) void pairwise_chunk_kernel(const uint8_t* __restrict__ src, float* __restrict__ sums, TableArg tab)
{
    __shared__ float s_tab[256];
    __shared__ float s_leaf[2][64];
    const int t = (int)threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) s_tab[t + 64 * i] = tab.f[t + 64 * i];
    __syncthreads();
    const uint8_t* p = src + (int64_t)blockIdx.x * kBuf + t * kLeaf;
    u32x4 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = ld16(p + 16 * i);
    float r[8];
#pragma unroll
    for (int it = 0; it < 16; ++it) {                       // elements 8 it .. 8 it + 7
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = 8 * it + j;
            const float v = s_tab[(q[e >> 4][(e >> 2) & 3] >> (8 * (e & 3))) & 255u];
            r[j] = it == 0 ? v : __fadd_rn(r[j], v);
        }
    }
    float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                          __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
    s_leaf[0][t] = res;
    __syncthreads();
    int cur = 0;
    for (int n = 32; n >= 1; n >>= 1) {                     // balanced tree: adjacent pairs, level by level
        if (t < n) s_leaf[cur ^ 1][t] = __fadd_rn(s_leaf[cur][2 * t], s_leaf[cur][2 * t + 1]);
        cur ^= 1;
        __syncthreads();
    }
    if (t == 0) sums[blockIdx.x] = s_leaf[cur][0];
}

// host: numpy's pairwise sum of tab[a[0..n)]
float pw_host(const uint8_t* a, int64_t n, const float* tab)
{
    if (n < 8) {
        float res = 0.f;
        for (int64_t i = 0; i < n; ++i) res = res + tab[a[i]];
        return res;
    }
    if (n <= kLeaf) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = tab[a[j]];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = r[j] + tab[a[i + j]];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + tab[a[i]];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    const float l = pw_host(a, n2, tab), r = pw_host(a + n2, n - n2, tab);
    return l + r;
}

}  // namespace

// np.add.reduce(table[src].astype(float32)) for a CONTIGUOUS uint8 array of `count` elements on the device, in numpy's
// order of additions.  table_host: 256 floats; *result_host receives the float32 sum.  BLOCKING: the stream is
// synchronised before returning (the value is needed on the host; server.py turns it into JSON).
extern "C" int ipx_pairwise_sum_lut_f32(void* stream, const uint8_t* src, int64_t count, const float* table_host, float* result_host)
{
    const char* fn = "ipx_pairwise_sum_lut_f32";
    ipx_clear_error();
    if (!src || !table_host || !result_host) return ipx_fail(IPX_ERR_INVALID, "%s: null pointer", fn);
    if (count <= 0) return ipx_fail(IPX_ERR_INVALID, "%s: count=%lld must be positive", fn, (long long)count);
    const int64_t nfull = count / kBuf, tail = count - nfull * kBuf;
    if (nfull > 0x7fffffff) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: array too large", fn);
    hipStream_t st = (hipStream_t)stream;
    std::vector<float> sums((size_t)nfull);
    std::vector<uint8_t> last((size_t)tail);
    float* dsums = nullptr;
    if (nfull) {
        IPX_HIP(hipMallocAsync((void**)&dsums, (size_t)nfull * sizeof(float), st));
        TableArg ta;
        __builtin_memcpy(ta.f, table_host, sizeof(ta.f));
        hipLaunchKernelGGL(pairwise_chunk_kernel, dim3((unsigned)nfull), dim3(64), 0, st, src, dsums, ta);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(sums.data(), dsums, (size_t)nfull * sizeof(float), hipMemcpyDeviceToHost, st);
        hipError_t e2 = hipFreeAsync(dsums, st);
        if (e != hipSuccess || e2 != hipSuccess) return ipx_fail(IPX_ERR_HIP, "%s: %s", fn, hipGetErrorString(e != hipSuccess ? e : e2));
    }
    if (tail) IPX_HIP(hipMemcpyAsync(last.data(), src + nfull * kBuf, (size_t)tail, hipMemcpyDeviceToHost, st));
    IPX_HIP(hipStreamSynchronize(st));
    float res = 0.f;                                        // the reduction starts from add's identity
    for (int64_t i = 0; i < nfull; ++i) res = res + sums[(size_t)i];
    if (tail) res = res + pw_host(last.data(), tail, table_host);
    *result_host = res;
    return IPX_OK;
}

// ipx_b64.hip -- base64.b64encode(buf) of the response's PNG (server.py:536, RFC 4648 standard alphabet, '=' padding).
// 40 ms of single-threaded host time per 4K colour response (binascii holds the GIL) against two PCIe copies and a kernel
// that moves 37 MB.  A thread turns 12 input bytes (three dwords) into 16 characters (one 16-byte store).
#include "ipx_common.h"

namespace {

__device__ __forceinline__ uint32_t b64_char(uint32_t x)              // 0..63 -> 'A'..'Z' 'a'..'z' '0'..'9' '+' '/'
{
    return x + 65u + (x >= 26u ? 6u : 0u) - (x >= 52u ? 75u : 0u) - (x >= 62u ? 15u : 0u) + (x >= 63u ? 3u : 0u);
}

__global__ __launch_bounds__(256) void base64_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int64_t n)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i0 = 12 * t;
    if (i0 >= n) return;
    uint32_t w[3] = {0u, 0u, 0u};
    const bool full = i0 + 12 <= n;
    if (full) {
        const uint32_t* p = (const uint32_t*)(src + i0);              // (src is 4-byte aligned: checked by the host)
        w[0] = p[0]; w[1] = p[1]; w[2] = p[2];
    } else {
        for (int i = 0; i0 + i < n; ++i) w[i >> 2] |= (uint32_t)src[i0 + i] << (8 * (i & 3));
    }
    uint32_t o[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        // bytes 3g, 3g+1, 3g+2 of the 12 (little-endian dwords)
        auto byte = [&](int i) { return (w[i >> 2] >> (8 * (i & 3))) & 255u; };
        const uint32_t v = (byte(3 * g) << 16) | (byte(3 * g + 1) << 8) | byte(3 * g + 2);
        o[g] = b64_char(v >> 18) | (b64_char((v >> 12) & 63u) << 8) | (b64_char((v >> 6) & 63u) << 16) | (b64_char(v & 63u) << 24);
    }
    uint8_t* d = dst + 16 * t;
    if (full) {
        u32x4 v = {o[0], o[1], o[2], o[3]};
        *(u32x4*)d = v;                                               // (dst + 16 t: 16-byte aligned, checked by the host)
    } else {
        const int rem = (int)(n - i0);                                // 1..11 bytes: 4 * ceil(rem / 3) characters, '=' padded
        const int nch = 4 * ((rem + 2) / 3), pad = (3 - rem % 3) % 3;
        for (int i = 0; i < nch; ++i) d[i] = i >= nch - pad ? (uint8_t)'=' : (uint8_t)(o[i >> 2] >> (8 * (i & 3)));
    }
}

}  // namespace

extern "C" int ipx_base64_encode_u8(void* stream, const uint8_t* src, int64_t n, uint8_t* dst)
{
    ipx_clear_error();
    if (n < 0) return ipx_fail(IPX_ERR_INVALID, "ipx_base64_encode_u8: n=%lld", (long long)n);
    if (n == 0) return IPX_OK;
    if (!src || !dst) return ipx_fail(IPX_ERR_INVALID, "ipx_base64_encode_u8: null pointer");
    if (((uintptr_t)src & 3u) || ((uintptr_t)dst & 15u))
        return ipx_fail(IPX_ERR_INVALID, "ipx_base64_encode_u8: src must be 4-byte and dst 16-byte aligned");
    const int64_t threads = (n + 11) / 12, blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffff) return ipx_fail(IPX_ERR_UNSUPPORTED, "ipx_base64_encode_u8: %lld bytes are too many", (long long)n);
    hipLaunchKernelGGL(base64_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    IPX_LAUNCH_CHECK("base64_kernel");
    return IPX_OK;
}

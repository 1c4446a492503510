// ipx_common.h -- shared host/device helpers of libipx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "ipx.h"

// ---- error reporting (thread-local, never throws across the C ABI) -------------------------------
int ipx_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void ipx_clear_error();

#define IPX_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return ipx_fail(IPX_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));   \
    } while (0)

#define IPX_LAUNCH_CHECK(name)                                                             \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess)                                                              \
            return ipx_fail(IPX_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

// Validates the common (n, h, w, c, strides) arguments of an image entry point.
// src_c / dst_c are the interleaved channel counts on either side.
int ipx_check_image_args(const char* fn, const void* src, const void* dst, int n, int h, int w,
                         int src_c, int dst_c, int64_t src_row_stride, int64_t src_img_stride,
                         int64_t dst_row_stride, int64_t dst_img_stride);

// ---- device helpers -------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// 16-byte loads/stores with NO alignment assumption: gfx950 global memory accepts unaligned vector
// accesses and hipcc lowers these to global_load/store_dwordx4 (checked in the .s).
__device__ __forceinline__ u32x4 ld16(const uint8_t* p) { u32x4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st16(uint8_t* p, u32x4 v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ u32x2 ld8(const uint8_t* p) { u32x2 v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ uint32_t ld4(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
__host__ __device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// XCD-aware work-item remap (guide T1, bijective form): blocks b and b+8 share an XCD (and its L2), so
// give each of the 8 residue classes a CONTIGUOUS range of work items.  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks)
{
    const unsigned q = nblocks >> 3, r = nblocks & 7u, x = bid & 7u;
    const unsigned base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

static inline unsigned ipx_div_up(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

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
// Neighbourhood operators read pixels around the one they write: a destination that overlaps the source would be read after
// it was overwritten (cv2 copies in that case; the reference never does it).  IPX_ERR_INVALID if the byte ranges
// [a, a + extent) and [b, b + extent) of two batches intersect; extent = (n-1) img_stride + (h-1) row_stride + row_bytes.
int ipx_check_disjoint(const char* fn, const char* what_a, const void* a, int64_t a_row_bytes, int64_t a_row, int64_t a_img,
                       const char* what_b, const void* b, int64_t b_row_bytes, int64_t b_row, int64_t b_img, int n, int h);
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

// Streaming ("nt") forms for data that is touched once: measured on MI355X, a block-contiguous copy with nt loads
// and stores runs 6.2-6.4 TB/s against 5.6 TB/s with the default cache policy (scripts/dev/membench.hip).
typedef u32x4 u32x4_u __attribute__((aligned(1)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_u __attribute__((aligned(1)));
typedef u32x2 u32x2_u __attribute__((aligned(1)));
typedef uint32_t u32_u __attribute__((aligned(1)));
__device__ __forceinline__ u32x4 ld16_nt(const uint8_t* p) { return __builtin_nontemporal_load((const u32x4_u*)p); }
__device__ __forceinline__ void st16_nt(uint8_t* p, u32x4 v) { __builtin_nontemporal_store(v, (u32x4_u*)p); }
template <bool NT> __device__ __forceinline__ u32x4 ld16x(const uint8_t* p) { return NT ? ld16_nt(p) : ld16(p); }
template <bool NT> __device__ __forceinline__ void st16x(uint8_t* p, u32x4 v) { if (NT) st16_nt(p, v); else st16(p, v); }

// Cache policy of the streaming kernels: a pass whose working set fits the 256 MiB Infinity Cache is left to the default
// policy (the next kernel of a pipeline finds its data there); larger ones stream with nt.  The crossovers were
// measured per call site on batches of 4K frames (scripts/dev/ntprobe.py), in bytes of source + destination touched:
//   table pass                    ~300 MB  (16 gray frames: default 2.58 vs nt 2.81 us/frame; 20 frames: 3.21 vs 2.78)
//   histogram -> table (gray)     ~460 MB  (the histogram pass should leave the image for the table pass: 24 frames
//                                           4.39 vs 4.62, 32 frames 4.71 vs 4.45)
//   BGR -> gray + histogram       ~420 MB of source (20 frames: 9.18 vs 8.88)
//   BGR -> gray                   always nt stores (never worse)
// IPX_NT_BYTES overrides every threshold (0 = always nt, huge = never).
//   histogram + tables only       ~600 MB  (2 x the gray batch; the caller's table pass comes next: 128 x 1080p gray, 265 MB, just
//                                           fits the 256 MiB cache: a six-table pass behind it 285 us (default) vs 420 us (nt))
constexpr uint64_t kNtLut = 300ull << 20, kNtHistThenLut = 440ull << 20, kNtGrayHist = 400ull << 20, kNtHistOnly = 600ull << 20, kNtAlways = 0;
bool ipx_use_nt(uint64_t bytes_touched, uint64_t site_threshold);

// gray = (3735 B + 19235 G + 9798 R + 16384) >> 15   (OpenCV RGB2Gray<uchar>, Q15)
__device__ __forceinline__ uint32_t gray_q15(uint32_t b, uint32_t g, uint32_t r)
{
    return (3735u * b + 19235u * g + 9798u * r + 16384u) >> 15;
}

template <int C> struct PxVec;                                   // 16 pixels of C interleaved bytes
template <> struct PxVec<3> { u32x4 q[3]; };
template <> struct PxVec<4> { u32x4 q[4]; };

template <int C, bool NT>
__device__ __forceinline__ void load_px16(PxVec<C>& p, const uint8_t* s)
{
#pragma unroll
    for (int i = 0; i < C; ++i) p.q[i] = ld16x<NT>(s + 16 * i);
}

// 1024 BGR pixels (3 KiB) of one wave, loaded as three CONTIGUOUS 1 KiB wave accesses (lane i takes bytes
// 1024 t + 16 i) and then transposed through a wave-private 3 KiB LDS region so that lane L ends up with its own 16
// pixels (bytes 48 L .. 48 L + 47).  Measured (scripts/dev/graybench.hip): with nt loads this streams at 6.3 TB/s,
// against 5.7 TB/s (default policy) / 5.0 TB/s (nt) when every lane loads its 48 bytes directly (48-byte lane stride).
// The b128 reads at a 48-byte lane stride are bank-conflict free.  No block barrier: the region belongs to one wave.
__device__ __forceinline__ void load_group3_nt(u32x4 (&v)[3], const uint8_t* group, int lane)
{
#pragma unroll
    for (int t = 0; t < 3; ++t) v[t] = ld16_nt(group + 1024 * t + 16 * lane);
}
__device__ __forceinline__ void transpose_group3(PxVec<3>& p, const u32x4 (&v)[3], uint8_t* wave_lds, int lane)
{
#pragma unroll
    for (int t = 0; t < 3; ++t) *reinterpret_cast<u32x4*>(wave_lds + 1024 * t + 16 * lane) = v[t];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < 3; ++t) p.q[t] = *reinterpret_cast<const u32x4*>(wave_lds + 48 * lane + 16 * t);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                       // the region may be rewritten after this point
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int C>
__device__ __forceinline__ uint32_t px_byte(const PxVec<C>& p, int byte_index)   // static after unrolling
{
    return (p.q[byte_index >> 4][(byte_index >> 2) & 3] >> (8 * (byte_index & 3))) & 255u;
}

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
__host__ __device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// Source row y of a block of h rows whose neighbours' rows may be supplied: `top` = the r rows above it (row -r first),
// `bot` = the r rows below it.  A side without halo rows is an image border: REFLECT_101 there -- which, for a block
// shorter than the radius, reaches past the block's far end into the OTHER side's halo rows.
__device__ __forceinline__ const uint8_t* halo_row(const uint8_t* simg, const uint8_t* top, const uint8_t* bot,
                                                   int y, int h, int r, int64_t s_row)
{
    if ((unsigned)y < (unsigned)h) return simg + (int64_t)y * s_row;
    if (y < 0 && !top) y = -y;
    else if (y >= h && !bot) y = 2 * (h - 1) - y;
    if (y < 0 && top) return top + (int64_t)(r + y) * s_row;
    if (y >= h && bot) return bot + (int64_t)(y - h) * s_row;
    return simg + (int64_t)reflect101(y, h) * s_row;
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

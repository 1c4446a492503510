// ipx_pointwise.hip -- pixel-wise kernels: 256-entry LUT application, BGR(A)->gray, fused gray+LUT.
//
// All are pure HBM streams (LUT: 2 B/px, gray: 4 B/px for 3 channels).  Each lane moves 16 output bytes
// per step with 16-byte global loads/stores; a block walks "units" of UNROLL x 4 KiB so that UNROLL
// independent loads are in flight per lane.  The table lives in LDS (256 B) and is read with ds_read_u8.
#include "ipx_common.h"
#include <stdlib.h>

namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = 4;
constexpr int kChunk = kThreads * 16;           // bytes (or pixels) one block-wide step covers
constexpr int kUnit = kChunk * kUnroll;

struct LutArg { uint32_t w[64]; };              // 256-byte table passed by value in the kernarg segment

struct PwGeom {
    int rows;                // rows per image after collapsing contiguous images into one long row
    int64_t len;             // OUTPUT elements per row (bytes for LUT, pixels for gray)
    int64_t s_row, s_img, d_row, d_img;
    unsigned units_per_row;  // ceil(len / kUnit)
    unsigned units_per_img;  // rows * units_per_row
};

__device__ __forceinline__ uint32_t lut4(const uint8_t* sl, uint32_t x)
{
    return (uint32_t)sl[x & 255u] | ((uint32_t)sl[(x >> 8) & 255u] << 8) |
           ((uint32_t)sl[(x >> 16) & 255u] << 16) | ((uint32_t)sl[x >> 24] << 24);
}

__device__ __forceinline__ void load_lut(uint32_t* s_lut, const LutArg& lutv, const uint8_t* lut_dev, bool byval)
{
    if (threadIdx.x < 64) {
        uint32_t v;
        if (byval) v = lutv.w[threadIdx.x];
        else __builtin_memcpy(&v, lut_dev + 4 * threadIdx.x, 4);
        s_lut[threadIdx.x] = v;
    }
    __syncthreads();
}

// dst = lut[src]; blockIdx.y = image, blockIdx.x grid-strides over the image's units.  The first unit's loads are
// issued BEFORE the table is fetched and staged, so that a device-resident table (equalizeHist) costs no extra
// round trip in these short-lived blocks.
template <bool BYVAL, bool NT>
__global__ __launch_bounds__(kThreads) void lut_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                       PwGeom g, LutArg lutv, const uint8_t* __restrict__ lut_dev,
                                                       int64_t lut_stride)
{
    __shared__ uint32_t s_lut[64];
    const unsigned img = blockIdx.y;
    const uint8_t* sl = reinterpret_cast<const uint8_t*>(s_lut);
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    bool have_lut = false;

    for (unsigned u = blockIdx.x; u < g.units_per_img; u += gridDim.x) {
        const unsigned row = u / g.units_per_row;
        const int64_t col0 = (int64_t)(u - row * g.units_per_row) * kUnit + (int64_t)threadIdx.x * 16;
        const uint8_t* sp = simg + (int64_t)row * g.s_row;
        uint8_t* dp = dimg + (int64_t)row * g.d_row;
        u32x4 v[kUnroll];
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            const int64_t c = col0 + (int64_t)j * kChunk;
            if (c + 16 <= g.len) v[j] = ld16x<NT>(sp + c);
        }
        if (!have_lut) {                                     // block-uniform
            load_lut(s_lut, lutv, BYVAL ? nullptr : lut_dev + (int64_t)img * lut_stride, BYVAL);
            have_lut = true;
        }
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            const int64_t c = col0 + (int64_t)j * kChunk;
            if (c + 16 <= g.len) {
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = lut4(sl, v[j][k]);
                st16x<NT>(dp + c, o);
            } else if (c < g.len) {                      // ragged tail of the row: byte by byte
                for (int64_t b = c; b < g.len; ++b) dp[b] = sl[sp[b]];
            }
        }
    }
}

// k tables applied to ONE read of the source: dst[j] = lut[j][src], j < k <= IPX_MAX_MULTI_LUT.  The pixel-wise modes of
// the reference (server.py:370-388, 459-467) all start from the same gray image; a batch pipeline that wants several of
// them pays 1 + k bytes per pixel here instead of 2 k.  Tables are per image (device memory, k x 256 bytes per image).
struct MultiDst { uint8_t* p[IPX_MAX_MULTI_LUT]; };

template <bool NT>
__global__ __launch_bounds__(kThreads) void lut_multi_kernel(const uint8_t* __restrict__ src, MultiDst dst, int k, PwGeom g,
                                                             const uint8_t* __restrict__ luts_dev, int64_t lut_img_stride)
{
    __shared__ uint32_t s_lut[IPX_MAX_MULTI_LUT * 64];
    const unsigned img = blockIdx.y;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    bool have_lut = false;
    for (unsigned u = blockIdx.x; u < g.units_per_img; u += gridDim.x) {
        const unsigned row = u / g.units_per_row;
        const int64_t col0 = (int64_t)(u - row * g.units_per_row) * kUnit + (int64_t)threadIdx.x * 16;
        const uint8_t* sp = simg + (int64_t)row * g.s_row;
        const int64_t doff = (int64_t)img * g.d_img + (int64_t)row * g.d_row;
        u32x4 v[kUnroll];
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            const int64_t c = col0 + (int64_t)j * kChunk;
            if (c + 16 <= g.len) v[j] = ld16x<NT>(sp + c);
        }
        if (!have_lut) {                                     // block-uniform; after the first unit's loads are in flight
            const uint8_t* lp = luts_dev + (int64_t)img * lut_img_stride;
            for (int i = (int)threadIdx.x; i < 64 * k; i += kThreads) {
                uint32_t w;
                __builtin_memcpy(&w, lp + 4 * i, 4);
                s_lut[i] = w;
            }
            __syncthreads();
            have_lut = true;
        }
        for (int t = 0; t < k; ++t) {
            const uint8_t* sl = reinterpret_cast<const uint8_t*>(s_lut) + 256 * t;
            uint8_t* dp = dst.p[t] + doff;
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) {
                const int64_t c = col0 + (int64_t)j * kChunk;
                if (c + 16 <= g.len) {
                    u32x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = lut4(sl, v[j][q]);
                    st16x<NT>(dp + c, o);
                } else if (c < g.len) {
                    for (int64_t b = c; b < g.len; ++b) dp[b] = sl[sp[b]];
                }
            }
        }
    }
}

// 16 gray (or table[gray]) bytes from a lane's 16 pixels
template <int C, bool WITH_LUT>
__device__ __forceinline__ u32x4 gray16(const PxVec<C>& p, const uint8_t* sl)
{
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t word = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = 4 * k + i;
            uint32_t y = gray_q15(px_byte<C>(p, px * C), px_byte<C>(p, px * C + 1), px_byte<C>(p, px * C + 2));
            if (WITH_LUT) y = sl[y];
            word |= y << (8 * i);
        }
        o[k] = word;
    }
    return o;
}

// dst = gray(src) or lut[gray(src)]; lane = 16 pixels (48 or 64 source bytes -> 16 output bytes).
// BGR: a wave whose 1024 pixels all lie inside the row takes them as three contiguous 1 KiB nt loads and sorts them
// out through LDS (transpose_group3); ragged groups and BGRA use per-lane loads.
template <int C, bool WITH_LUT, bool NT>
__global__ __launch_bounds__(kThreads) void gray_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                        PwGeom g, LutArg lutv)
{
    __shared__ uint32_t s_lut[64];
    __shared__ __attribute__((aligned(16))) uint8_t s_tr[C == 3 ? (kThreads / 64) * 3072 : 16];
    if (WITH_LUT) load_lut(s_lut, lutv, nullptr, true);
    const uint8_t* sl = reinterpret_cast<const uint8_t*>(s_lut);
    const unsigned img = blockIdx.y;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    constexpr int U = 2;                                  // 2 x (3|4) x 16 B loads in flight per lane
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = (int)(threadIdx.x >> 6);
    uint8_t* wlds = s_tr + (C == 3 ? wave * 3072 : 0);

    for (unsigned u = blockIdx.x; u < g.units_per_img; u += gridDim.x) {
        const unsigned row = u / g.units_per_row;
        const int64_t gx0 = (int64_t)(u - row * g.units_per_row) * (kChunk * U) + (int64_t)wave * 1024;   // wave's group
        const uint8_t* sp = simg + (int64_t)row * g.s_row;
        uint8_t* dp = dimg + (int64_t)row * g.d_row;
        PxVec<C> p[U];
        u32x4 raw[U][3];
        bool full[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int64_t gx = gx0 + (int64_t)j * kChunk, x = gx + 16 * lane;
            full[j] = C == 3 && gx + 1024 <= g.len;       // wave-uniform
            if (full[j]) load_group3_nt(raw[j], sp + gx * 3, lane);
            else if (x + 16 <= g.len) load_px16<C, NT>(p[j], sp + x * C);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int64_t x = gx0 + (int64_t)j * kChunk + 16 * lane;
            if (C == 3 && full[j]) {
                PxVec<3> q;
                transpose_group3(q, raw[j], wlds, lane);
                __builtin_memcpy(&p[j], &q, sizeof(q));
            }
            if (x + 16 <= g.len) {
                st16x<NT>(dp + x, gray16<C, WITH_LUT>(p[j], sl));
            } else if (x < g.len) {
                for (int64_t q = x; q < g.len; ++q) {
                    const uint8_t* s = sp + q * C;
                    uint32_t y = gray_q15(s[0], s[1], s[2]);
                    if (WITH_LUT) y = sl[y];
                    dp[q] = (uint8_t)y;
                }
            }
        }
    }
}

// len_in_row: output elements per row; in_bpe: source bytes per output element
PwGeom make_geom(int h, int64_t len_in_row, int in_bpe, int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                 int unit)
{
    PwGeom g;
    g.rows = h; g.len = len_in_row;
    g.s_row = s_row; g.s_img = s_img; g.d_row = d_row; g.d_img = d_img;
    if (s_row == len_in_row * in_bpe && d_row == len_in_row) {   // contiguous image: one long row
        g.rows = 1; g.len = len_in_row * (int64_t)h;
    }
    g.units_per_row = ipx_div_up((uint64_t)g.len, (uint64_t)unit);
    g.units_per_img = (unsigned)g.rows * g.units_per_row;
    return g;
}

dim3 make_grid(const PwGeom& g, int n)
{
    // One contiguous unit per block and as many blocks as there are units: on MI355X that streams faster than a
    // capped grid that strides over the image (6.2 vs 5.6 TB/s on a 400 MiB copy, scripts/dev/membench.hip).
    unsigned per_img = g.units_per_img;
    if (per_img > 0x7fffffffu) per_img = 0x7fffffffu;         // the kernels still stride if ever needed
    return dim3(per_img, (unsigned)n, 1);
}

int check_units(const char* fn, const PwGeom& g, int h, int64_t len, int unit)
{
    const uint64_t upr = ((uint64_t)g.len + unit - 1) / unit;
    if (upr * (uint64_t)g.rows > 0xffffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: image too large", fn);
    (void)h; (void)len;
    return IPX_OK;
}

// (Tried and removed, round 2: walking the batch from its END for the table pass of equalizeHist, so that its first reads hit
// what the histogram pass left in the 256 MiB Infinity Cache -- no measurable difference.)
int launch_lut(const char* fn, void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
               int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
               const uint8_t* lut_host, const uint8_t* lut_dev, int64_t lut_stride)
{
    if (int rc = ipx_check_image_args(fn, src, dst, n, h, w, c, c, s_row, s_img, d_row, d_img)) return rc;
    PwGeom g = make_geom(h, (int64_t)w * c, 1, s_row, s_img, d_row, d_img, kUnit);
    if (int rc = check_units(fn, g, h, (int64_t)w * c, kUnit)) return rc;
    hipStream_t st = (hipStream_t)stream;
    LutArg la{};
    if (lut_host) __builtin_memcpy(la.w, lut_host, 256);
    const bool nt = ipx_use_nt(2ull * (uint64_t)n * h * w * c, kNtLut);
    const int nchunks = (n + 65534) / 65535;
    for (int ci = 0; ci < nchunks; ++ci) {
        const int i0 = ci * 65535;
        const int nb = n - i0 < 65535 ? n - i0 : 65535;
        const uint8_t* s = src + (int64_t)i0 * s_img;
        uint8_t* d = dst + (int64_t)i0 * d_img;
        dim3 grid = make_grid(g, nb);
        const uint8_t* ld = lut_host ? nullptr : lut_dev + (int64_t)i0 * lut_stride;
        auto k = lut_host ? (nt ? lut_kernel<true, true> : lut_kernel<true, false>)
                          : (nt ? lut_kernel<false, true> : lut_kernel<false, false>);
        hipLaunchKernelGGL(k, grid, dim3(kThreads), 0, st, s, d, g, la, ld, lut_stride);
        IPX_LAUNCH_CHECK("lut_kernel");
    }
    return IPX_OK;
}

int launch_gray(const char* fn, void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int sc,
                int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img, const uint8_t* lut_host)
{
    if (sc != 3 && sc != 4) {
        ipx_clear_error();
        return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: src_channels=%d (3 = BGR or 4 = BGRA expected)", fn, sc);
    }
    if (int rc = ipx_check_image_args(fn, src, dst, n, h, w, sc, 1, s_row, s_img, d_row, d_img)) return rc;
    constexpr int unit = kChunk * 2;
    PwGeom g = make_geom(h, (int64_t)w, sc, s_row, s_img, d_row, d_img, unit);
    if (int rc = check_units(fn, g, h, w, unit)) return rc;
    hipStream_t st = (hipStream_t)stream;
    LutArg la{};
    if (lut_host) __builtin_memcpy(la.w, lut_host, 256);
    const bool nt = ipx_use_nt((uint64_t)n * h * w * (sc + 1), kNtAlways);
    for (int i0 = 0; i0 < n; i0 += 65535) {
        const int nb = n - i0 < 65535 ? n - i0 : 65535;
        const uint8_t* s = src + (int64_t)i0 * s_img;
        uint8_t* d = dst + (int64_t)i0 * d_img;
        dim3 grid = make_grid(g, nb);
        auto k = sc == 3 ? (lut_host ? (nt ? gray_kernel<3, true, true> : gray_kernel<3, true, false>)
                                     : (nt ? gray_kernel<3, false, true> : gray_kernel<3, false, false>))
                         : (lut_host ? (nt ? gray_kernel<4, true, true> : gray_kernel<4, true, false>)
                                     : (nt ? gray_kernel<4, false, true> : gray_kernel<4, false, false>));
        hipLaunchKernelGGL(k, grid, dim3(kThreads), 0, st, s, d, g, la);
        IPX_LAUNCH_CHECK("gray_kernel");
    }
    return IPX_OK;
}

// 16-bit sources (server.py:516-523): cv2.cvtColor(img, COLOR_BGR2GRAY) on uint16 -- OpenCV's RGB2Gray<ushort>, the same Q15
// coefficients as for bytes: gray16 = (3735 B + 19235 G + 9798 R + 16384) >> 15 (at most 65535 * 32768 + 16384 < 2^31) -- then
// (img >> 8).astype(uint8).  C = samples per pixel: 1 and 2 (gray, gray + alpha) only shift, 3 and 4 convert (alpha dropped).
// BE: samples big-endian, as a PNG stores them; RGB: R first (file order) instead of OpenCV's B first.  A thread takes 8 pixels:
// C 16-byte loads, one 8-byte store.
template <int C, bool BE, bool RGB>
__device__ __forceinline__ uint32_t gray16_px(const uint32_t (&w)[4 * C], int px)           // static indices after unrolling
{
    auto sample = [&](int idx) -> uint32_t {
        uint32_t v = (w[idx >> 1] >> (16 * (idx & 1))) & 0xffffu;
        if (BE) v = ((v & 0xffu) << 8) | (v >> 8);
        return v;
    };
    if (C < 3) return sample(px * C) >> 8;
    const uint32_t c0 = sample(px * C), c1 = sample(px * C + 1), c2 = sample(px * C + 2);
    return gray_q15(RGB ? c2 : c0, c1, RGB ? c0 : c2) >> 8;
}

template <int C, bool BE, bool RGB>
__global__ __launch_bounds__(kThreads) void gray16_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, PwGeom g)
{
    const unsigned img = blockIdx.y;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    for (unsigned u = blockIdx.x; u < g.units_per_img; u += gridDim.x) {
        const unsigned row = u / g.units_per_row;
        const int64_t x = (int64_t)(u - row * g.units_per_row) * (kThreads * 8) + (int64_t)threadIdx.x * 8;
        const uint8_t* sp = simg + (int64_t)row * g.s_row;
        uint8_t* dp = dimg + (int64_t)row * g.d_row;
        if (x + 8 <= g.len) {
            uint32_t w[4 * C];
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const u32x4 v = ld16_nt(sp + x * (2 * C) + 16 * i);
#pragma unroll
                for (int q = 0; q < 4; ++q) w[4 * i + q] = v[q];
            }
            u32x2 o = {0u, 0u};
#pragma unroll
            for (int px = 0; px < 8; ++px) o[px >> 2] |= gray16_px<C, BE, RGB>(w, px) << (8 * (px & 3));
            __builtin_nontemporal_store(o, (u32x2_u*)(dp + x));
        } else {
            for (int64_t q = x; q < g.len; ++q) {
                const uint8_t* s = sp + q * (2 * C);
                auto sample = [&](int ch) -> uint32_t {
                    return BE ? ((uint32_t)s[2 * ch] << 8 | s[2 * ch + 1]) : ((uint32_t)s[2 * ch + 1] << 8 | s[2 * ch]);
                };
                dp[q] = (uint8_t)(C < 3 ? sample(0) >> 8 : gray_q15(sample(RGB ? 2 : 0), sample(1), sample(RGB ? 0 : 2)) >> 8);
            }
        }
    }
}

template <int C>
void launch_gray16_c(hipStream_t st, dim3 grid, const uint8_t* s, uint8_t* d, const PwGeom& g, bool be, bool rgb)
{
    auto k = be ? (rgb ? gray16_kernel<C, true, true> : gray16_kernel<C, true, false>)
                : (rgb ? gray16_kernel<C, false, true> : gray16_kernel<C, false, false>);
    hipLaunchKernelGGL(k, grid, dim3(kThreads), 0, st, s, d, g);
}

}  // namespace

extern "C" {

int ipx_lut_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
               int64_t src_row_stride, int64_t src_img_stride, int64_t dst_row_stride, int64_t dst_img_stride,
               const uint8_t* lut_host)
{
    if (!lut_host) { ipx_clear_error(); return ipx_fail(IPX_ERR_INVALID, "ipx_lut_u8: null table"); }
    return launch_lut("ipx_lut_u8", stream, src, dst, n, h, w, c, src_row_stride, src_img_stride,
                      dst_row_stride, dst_img_stride, lut_host, nullptr, 0);
}

int ipx_lut_dev_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                   int64_t src_row_stride, int64_t src_img_stride, int64_t dst_row_stride, int64_t dst_img_stride,
                   const uint8_t* lut_dev, int64_t lut_stride)
{
    if (!lut_dev) { ipx_clear_error(); return ipx_fail(IPX_ERR_INVALID, "ipx_lut_dev_u8: null table"); }
    if (lut_stride != 0 && lut_stride < 256) { ipx_clear_error(); return ipx_fail(IPX_ERR_INVALID, "ipx_lut_dev_u8: lut_stride %lld must be 0 or >= 256", (long long)lut_stride); }
    return launch_lut("ipx_lut_dev_u8", stream, src, dst, n, h, w, c, src_row_stride, src_img_stride,
                      dst_row_stride, dst_img_stride, nullptr, lut_dev, lut_stride);
}

int ipx_lut_multi_u8(void* stream, const uint8_t* src, uint8_t* const* dsts, int k, int n, int h, int w, int c,
                     int64_t src_row_stride, int64_t src_img_stride, int64_t dst_row_stride, int64_t dst_img_stride,
                     const uint8_t* luts_dev, int64_t lut_img_stride)
{
    const char* fn = "ipx_lut_multi_u8";
    ipx_clear_error();
    if (k < 1 || k > IPX_MAX_MULTI_LUT) return ipx_fail(IPX_ERR_INVALID, "%s: k=%d must be in 1..%d", fn, k, IPX_MAX_MULTI_LUT);
    if (!dsts || !luts_dev) return ipx_fail(IPX_ERR_INVALID, "%s: null %s", fn, dsts ? "tables" : "destination list");
    if (lut_img_stride != 0 && lut_img_stride < 256 * (int64_t)k)
        return ipx_fail(IPX_ERR_INVALID, "%s: lut_img_stride %lld must be 0 or >= %d", fn, (long long)lut_img_stride, 256 * k);
    MultiDst md{};
    for (int j = 0; j < k; ++j) {
        if (int rc = ipx_check_image_args(fn, src, dsts[j], n, h, w, c, c, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride)) return rc;
        for (int i = 0; i < j; ++i)
            if (dsts[i] == dsts[j]) return ipx_fail(IPX_ERR_INVALID, "%s: destinations %d and %d are the same buffer", fn, i, j);
        // with k > 1 every table reads the source after destination j has been written: no destination may overlap it
        // (k = 1 in place is the plain table pass and stays allowed)
        if (k > 1) {
            const int64_t s_span = (int64_t)(n - 1) * src_img_stride + (int64_t)(h - 1) * src_row_stride + (int64_t)w * c;
            const int64_t d_span = (int64_t)(n - 1) * dst_img_stride + (int64_t)(h - 1) * dst_row_stride + (int64_t)w * c;
            if (dsts[j] < src + s_span && src < dsts[j] + d_span)
                return ipx_fail(IPX_ERR_INVALID, "%s: destination %d overlaps the source (allowed only for k = 1)", fn, j);
        }
        md.p[j] = dsts[j];
    }
    PwGeom g = make_geom(h, (int64_t)w * c, 1, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride, kUnit);
    if (int rc = check_units(fn, g, h, (int64_t)w * c, kUnit)) return rc;
    const bool nt = ipx_use_nt((uint64_t)(1 + k) * n * h * w * c, kNtLut);
    for (int i0 = 0; i0 < n; i0 += 65535) {
        const int nb = n - i0 < 65535 ? n - i0 : 65535;
        MultiDst d = md;
        for (int j = 0; j < k; ++j) d.p[j] += (int64_t)i0 * dst_img_stride;
        hipLaunchKernelGGL(nt ? lut_multi_kernel<true> : lut_multi_kernel<false>, make_grid(g, nb), dim3(kThreads), 0, (hipStream_t)stream,
                           src + (int64_t)i0 * src_img_stride, d, k, g, luts_dev + (int64_t)i0 * lut_img_stride, lut_img_stride);
        IPX_LAUNCH_CHECK("lut_multi_kernel");
    }
    return IPX_OK;
}

int ipx_bgr2gray_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int src_channels,
                    int64_t src_row_stride, int64_t src_img_stride, int64_t dst_row_stride, int64_t dst_img_stride)
{
    return launch_gray("ipx_bgr2gray_u8", stream, src, dst, n, h, w, src_channels, src_row_stride, src_img_stride,
                       dst_row_stride, dst_img_stride, nullptr);
}

int ipx_bgr2gray_lut_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int src_channels,
                        int64_t src_row_stride, int64_t src_img_stride, int64_t dst_row_stride, int64_t dst_img_stride,
                        const uint8_t* lut_host)
{
    if (!lut_host) { ipx_clear_error(); return ipx_fail(IPX_ERR_INVALID, "ipx_bgr2gray_lut_u8: null table"); }
    return launch_gray("ipx_bgr2gray_lut_u8", stream, src, dst, n, h, w, src_channels, src_row_stride, src_img_stride,
                       dst_row_stride, dst_img_stride, lut_host);
}

int ipx_gray_from_u16_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int src_channels, int flags,
                         int64_t src_row_stride, int64_t src_img_stride, int64_t dst_row_stride, int64_t dst_img_stride)
{
    const char* fn = "ipx_gray_from_u16_u8";
    ipx_clear_error();
    if (src_channels < 1 || src_channels > 4)
        return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: src_channels=%d (1..4 16-bit samples per pixel expected)", fn, src_channels);
    if (flags & ~(IPX_U16_BIG_ENDIAN | IPX_U16_RGB_ORDER)) return ipx_fail(IPX_ERR_INVALID, "%s: unknown flags 0x%x", fn, (unsigned)flags);
    if (int rc = ipx_check_image_args(fn, src, dst, n, h, w, 1, 1, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride)) return rc;
    const int64_t row_bytes = (int64_t)w * 2 * src_channels;                     // (the helper above knows 1..4 BYTES per pixel)
    if (src_row_stride < row_bytes) return ipx_fail(IPX_ERR_INVALID, "%s: src_row_stride %lld < %lld row bytes", fn, (long long)src_row_stride, (long long)row_bytes);
    if (n > 1 && src_img_stride < src_row_stride * (int64_t)(h - 1) + row_bytes)
        return ipx_fail(IPX_ERR_INVALID, "%s: src_img_stride %lld smaller than one image", fn, (long long)src_img_stride);
    constexpr int unit = kThreads * 8;
    PwGeom g = make_geom(h, (int64_t)w, 2 * src_channels, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride, unit);
    if (int rc = check_units(fn, g, h, w, unit)) return rc;
    const bool be = (flags & IPX_U16_BIG_ENDIAN) != 0, rgb = (flags & IPX_U16_RGB_ORDER) != 0;
    for (int i0 = 0; i0 < n; i0 += 65535) {
        const int nb = n - i0 < 65535 ? n - i0 : 65535;
        const uint8_t* s = src + (int64_t)i0 * src_img_stride;
        uint8_t* d = dst + (int64_t)i0 * dst_img_stride;
        const dim3 grid = make_grid(g, nb);
        switch (src_channels) {
        case 1: launch_gray16_c<1>((hipStream_t)stream, grid, s, d, g, be, rgb); break;
        case 2: launch_gray16_c<2>((hipStream_t)stream, grid, s, d, g, be, rgb); break;
        case 3: launch_gray16_c<3>((hipStream_t)stream, grid, s, d, g, be, rgb); break;
        default: launch_gray16_c<4>((hipStream_t)stream, grid, s, d, g, be, rgb); break;
        }
        IPX_LAUNCH_CHECK("gray16_kernel");
    }
    return IPX_OK;
}

}  // extern "C"

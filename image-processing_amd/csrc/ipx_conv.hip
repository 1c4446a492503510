// ipx_conv.hip -- 3x3 convolutions (Laplacian sharpen, box, Sobel) and the separable fixed-point
// Gaussian of cv2.GaussianBlur, for 1..4 interleaved uint8 channels, BORDER_REFLECT_101.
//
// Work decomposition ("strip walker"): an image row is L = w*c bytes.  One WAVE owns a strip of up to 64 lanes x
// 16 bytes of the row and walks down a segment of rows.  Per input row a lane loads its own 16 bytes plus the
// few halo bytes left and right of them (the cache lines its neighbour lanes load: L1 hits), turns them into
// packed 16-bit "field pairs" (v_perm_b32) and does the horizontal part of the filter with packed 16-bit VALU ops;
// the vertical window of those row features lives in registers.  Loads run U rows ahead of the arithmetic.
// Every input byte is fetched from HBM once (plus r halo rows per segment); nothing is staged twice.
// A row of L bytes is cut into ceil(L / 16) lanes; the last lane is shifted left so that it ends at the row's last byte
// (and the one before it so that its halo stays inside the row): lanes may overlap there and store the same bytes
// twice.  The first and last lane build their missing halo bytes in registers (REFLECT_101 in pixel space,
// ipx_fields.h fix_edges), so the whole image is computed by the vector lanes -- no separate pass over border bytes,
// which used to be two thirds of the waves of a 4K gray launch and its tail.  Only images whose rows are shorter than
// 32 + halo bytes are computed per byte ("border" blocks: one byte per thread, plain code).
// Neighbouring strips are consecutive work items and the block->item map is XCD-aware so that they share an L2.
// 9..31 taps: ipx_sepmfma.hip (both passes as banded int8 matrix products on the MFMA units; taps above 127 fall back
// to ipx_sepbig.hip, a row-pass kernel + column-pass kernel around a Q8.8 scratch image).  Anything larger
// goes through the simple generic two-pass path at the bottom of this file; k = 1 is a copy.
#include "ipx_common.h"
#include <stdio.h>
#include <stdlib.h>
#include "ipx_fields.h"
#include <type_traits>
#include <atomic>

extern "C" int ipx_lut_u8(void*, const uint8_t*, uint8_t*, int, int, int, int, int64_t, int64_t, int64_t, int64_t, const uint8_t*);

int ipx_sep_mfma(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                 int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                 const uint16_t* kx, const uint16_t* ky, int k, const uint8_t* top, const uint8_t* bot);

namespace {

constexpr int kMaxWavesPerBlock = 16;
#ifndef IPX_CONV3_U
#define IPX_CONV3_U 4
#endif
constexpr int kConv3U = IPX_CONV3_U;


struct ConvGeom {
    int h, w, L;                       // rows, pixels per row, bytes per row
    int64_t s_row, s_img, d_row, d_img;
    int nlanes;                        // 16-byte lanes per row (0: rows too short, the whole image goes per byte)
    int lanes_per_strip, nstrips, seg_rows, nsegs;
    int wpb, blocks_per_seg;           // waves per block, blocks per (image, segment)
    unsigned main_blocks;              // blocks [0, main_blocks) walk strips, the rest compute single bytes
    int nb;                            // bytes per row computed singly: 0, or all of L for short rows
    unsigned border_total;             // n * h * nb
    const uint8_t* top;                // r halo rows above row 0 (row -r first), or null = reflect
    const uint8_t* bot;                // r halo rows below row h-1, or null = reflect
#ifdef IPX_DEV
    unsigned long long* trace;         // dev build (IPX_CONV_TRACE): per wave start / end time stamps, else null
#endif
};

// Source row `y` of the image (y may lie outside [0,h)): halo row if provided, else REFLECT_101.
__device__ __forceinline__ const uint8_t* row_ptr(const ConvGeom& g, const uint8_t* simg, int y, int r)
{
    return halo_row(simg, g.top, g.bot, y, g.h, r, g.s_row);
}

__device__ __forceinline__ ss2 pk_max(ss2 a, ss2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ ss2 pk_min(ss2 a, ss2 b) { return __builtin_elementwise_min(a, b); }

enum { OP_SHARPEN = 0, OP_BOX = 1, OP_SOBEL = 2 };

// ---- 3x3 operators as (row feature, vertical combine) pairs ------------------------------------------------
template <int OP, int C> struct Conv3;

template <int C> struct Conv3<OP_SHARPEN, C> {          // clip(5c - n - s - e - w)        server.py:270-276
    struct Feat { us2 c[8]; ss2 t[8]; };               // centre fields; t = 5c - (e + w)
    static __device__ __forceinline__ void feat(Feat& f, const Win<C, 1>& W)
    {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int p = slot_p(s);
            f.c[s] = fld<C, 1>(W, p);
            const us2 hs = fld<C, 1>(W, p - C) + fld<C, 1>(W, p + C);
            f.t[s] = __builtin_bit_cast(ss2, (us2)(f.c[s] * (unsigned short)5 - hs));
        }
    }
    static __device__ __forceinline__ u32x4 out(const Feat& a, const Feat& b, const Feat& c)
    {
        u32x4 o;
        us2 r[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            ss2 v = b.t[s] - __builtin_bit_cast(ss2, (us2)(a.c[s] + c.c[s]));
            v = pk_min(pk_max(v, (ss2)(0)), (ss2)(255));
            r[s] = __builtin_bit_cast(us2, v);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack_fields(r[2 * k], r[2 * k + 1]);
        return o;
    }
};

template <int C> struct Conv3<OP_BOX, C> {              // round(sum9 / 9): never a tie -> (sum + 4) / 9
    struct Feat { us2 s3[8]; };
    static __device__ __forceinline__ void feat(Feat& f, const Win<C, 1>& W)
    {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int p = slot_p(s);
            f.s3[s] = fld<C, 1>(W, p - C) + fld<C, 1>(W, p) + fld<C, 1>(W, p + C);
        }
    }
    static __device__ __forceinline__ u32x4 out(const Feat& a, const Feat& b, const Feat& c)
    {
        u32x4 o;
        uint32_t lo[8], hi[8];
        const uint32_t m = __builtin_amdgcn_readfirstlane(7282u);      // (x * 7282) >> 16 == x / 9 for x < 3640*9
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const uint32_t sum = __builtin_bit_cast(uint32_t, (us2)(a.s3[s] + b.s3[s] + c.s3[s]));
            lo[s] = mad_lo16(sum, m, 4u * 7282u);
            hi[s] = mad_hi16(sum, m, 4u * 7282u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack_acc(lo[2 * k], lo[2 * k + 1], hi[2 * k], hi[2 * k + 1]);
        return o;
    }
};

template <int C> struct Conv3<OP_SOBEL, C> {            // min(255, |gx| + |gy|)
    struct Feat { ss2 d[8]; us2 s[8]; };               // d = right - left, s = left + 2 centre + right
    static __device__ __forceinline__ void feat(Feat& f, const Win<C, 1>& W)
    {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int p = slot_p(s);
            const us2 l = fld<C, 1>(W, p - C), m = fld<C, 1>(W, p), r = fld<C, 1>(W, p + C);
            f.d[s] = __builtin_bit_cast(ss2, (us2)(r - l));
            f.s[s] = l + r + m * (unsigned short)2;
        }
    }
    static __device__ __forceinline__ u32x4 out(const Feat& a, const Feat& b, const Feat& c)
    {
        u32x4 o;
        us2 r[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const ss2 gx = a.d[s] + c.d[s] + b.d[s] * (short)2;
            const ss2 gy = __builtin_bit_cast(ss2, (us2)(c.s[s] - a.s[s]));
            const ss2 m = pk_max(gx, -gx) + pk_max(gy, -gy);            // <= 2040, no overflow
            r[s] = __builtin_bit_cast(us2, pk_min(m, (ss2)(255)));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack_fields(r[2 * k], r[2 * k + 1]);
        return o;
    }
};

// Per-byte evaluation of the same operators (border bytes, tiny images).
template <int OP>
__device__ __forceinline__ uint32_t conv3_byte(const uint8_t* r0, const uint8_t* r1, const uint8_t* r2, int xl, int xc, int xr)
{
    const int a = r0[xl], b = r0[xc], c = r0[xr], d = r1[xl], e = r1[xc], f = r1[xr], g = r2[xl], h = r2[xc], i = r2[xr];
    int v;
    if (OP == OP_SHARPEN) { v = 5 * e - (b + h + d + f); v = v < 0 ? 0 : v > 255 ? 255 : v; }
    else if (OP == OP_BOX) { v = (a + b + c + d + e + f + g + h + i + 4) / 9; }
    else {
        const int gx = (c + 2 * f + i) - (a + 2 * d + g), gy = (g + 2 * h + i) - (a + 2 * b + c);
        v = (gx < 0 ? -gx : gx) + (gy < 0 ? -gy : gy);
        if (v > 255) v = 255;
    }
    return (uint32_t)v;
}

// Decode a border thread: false if out of range.  b = byte in the row, y = row, img.
__device__ __forceinline__ bool border_item(const ConvGeom& g, unsigned bblock, unsigned& img, int& y, int& b)
{
    const unsigned t = bblock * blockDim.x + threadIdx.x;
    if (t >= g.border_total) return false;
    const unsigned j = t % (unsigned)g.nb, rest = t / (unsigned)g.nb;
    y = (int)(rest % (unsigned)g.h);
    img = rest / (unsigned)g.h;
    b = (int)j;
    return true;
}

// Decode the wave's strip item; all values wave-uniform (SGPRs).  A block of wpb = blockDim.x / 64 waves covers wpb
// ADJACENT strips of one row segment, so that a block streams a wide contiguous span of every row it walks.
__device__ __forceinline__ bool wave_item(const ConvGeom& g, unsigned& img, int& y0, int& y1, int& lane0)
{
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned blk = xcd_remap(blockIdx.x, g.main_blocks);
    const unsigned sub = blk % (unsigned)g.blocks_per_seg;
    const unsigned rest = blk / (unsigned)g.blocks_per_seg;
    const unsigned strip = sub * (unsigned)g.wpb + wave;
    if (strip >= (unsigned)g.nstrips) return false;
    const unsigned seg = rest % (unsigned)g.nsegs;
    img = rest / (unsigned)g.nsegs;
    y0 = (int)seg * g.seg_rows;
    y1 = y0 + g.seg_rows < g.h ? y0 + g.seg_rows : g.h;
    lane0 = (int)strip * g.lanes_per_strip;
    return true;
}

// First row byte of 16-byte lane gl (0 <= gl < nlanes): 16 gl, except that the last lane ends at the row's last byte and
// the one before it ends where the last one starts at the latest (its right halo then lies inside the row).
__device__ __forceinline__ int lane_b0(const ConvGeom& g, int gl)
{
    int b0 = 16 * gl;
    if (gl == g.nlanes - 1) b0 = g.L - 16;
    else if (gl == g.nlanes - 2 && b0 > g.L - 32) b0 = g.L - 32;
    return b0;
}

template <int OP, int C>
__global__ __launch_bounds__(64 * kMaxWavesPerBlock) void conv3_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                         ConvGeom g)
{
    if (blockIdx.x >= g.main_blocks) {                       // ---- border bytes, one per thread
        unsigned img; int y, b;
        if (!border_item(g, blockIdx.x - g.main_blocks, img, y, b)) return;
        const uint8_t* simg = src + (int64_t)img * g.s_img;
        const int px = b / C, ch = b - px * C;
        const int xl = reflect101(px - 1, g.w) * C + ch, xr = reflect101(px + 1, g.w) * C + ch;
        dst[(int64_t)img * g.d_img + (int64_t)y * g.d_row + b] = (uint8_t)conv3_byte<OP>(
            row_ptr(g, simg, y - 1, 1), row_ptr(g, simg, y, 1), row_ptr(g, simg, y + 1, 1), xl, b, xr);
        return;
    }
    unsigned img; int y0, y1, lane0;
    if (!wave_item(g, img, y0, y1, lane0)) return;
    const int gl = lane0 + (int)(threadIdx.x & 63u);         // 16-byte lane of the row
    if (gl >= g.nlanes) return;
    const int b0 = lane_b0(g, gl);
    const bool first = b0 == 0, last = b0 + 16 == g.L;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    typedef Win<C, 1> W1;
    typedef Conv3<OP, C> Op;
    constexpr int U = kConv3U;                               // rows per step; loads run one step ahead

    // (Walking odd segments upwards so that neighbours meet at their shared halo rows was tried: no gain, removed.)
    const int nrows = y1 - y0;
    constexpr int dir = 1;
    const int ys = y0;                                       // first output row
    auto row = [&](int t) {                                  // t-th source row of the walk, t = 0 .. nrows + 1
#if IPX_ABLATE & 2          // dev ablation: clamped rows instead of halo rows / REFLECT_101 (timing only)
        int yy = ys + (t <= nrows + 1 ? t : nrows + 1) - 1;
        yy = yy < 0 ? 0 : yy > g.h - 1 ? g.h - 1 : yy;
        return simg + (int64_t)yy * g.s_row;
#else
        return row_ptr(g, simg, ys + dir * ((t <= nrows + 1 ? t : nrows + 1) - 1), 1);
#endif
    };
    typename Op::Feat fa, fb;
    {
        W1 wa, wb;
        load_win<C, 1>(wa, row(0), b0, g.L);
        load_win<C, 1>(wb, row(1), b0, g.L);
        fix_edges<C, 1>(wa, first, last);
        fix_edges<C, 1>(wb, first, last);
        Op::feat(fa, wa);
        Op::feat(fb, wb);
    }
    W1 nx[U];
#pragma unroll
    for (int j = 0; j < U; ++j) load_win<C, 1>(nx[j], row(2 + j), b0, g.L);
    for (int i = 0; i < nrows; i += U) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            typename Op::Feat fc;
            fix_edges<C, 1>(nx[j], first, last);
            Op::feat(fc, nx[j]);
#if IPX_ABLATE & 4          // dev ablation: no arithmetic, the window's own 16 bytes are stored (timing only)
            u32x4 ctr;
            __builtin_memcpy(&ctr, &nx[j].w[W1::HD], 16);
            ctr[0] ^= nx[j].w[0] ^ nx[j].w[W1::HD + 4];
#endif
            load_win<C, 1>(nx[j], row(i + U + 2 + j), b0, g.L);    // the slot is free: refill it with the row U steps further on
#if IPX_ABLATE & 4
            if (i + j < nrows) st16_nt(dimg + (int64_t)(ys + dir * (i + j)) * g.d_row + b0, ctr);
#else
            if (i + j < nrows) st16_nt(dimg + (int64_t)(ys + dir * (i + j)) * g.d_row + b0, Op::out(fa, fb, fc));
#endif
            fa = fb;
            fb = fc;
        }
    }
}

// The same walker with a SECOND output from the same read: dst2 = table[src] (per-image tables in device memory).  A batch
// pipeline that wants a 3 x 3 operator AND a pixel-wise table pass of one gray batch -- BASELINE.json configs[1]: Sobel of the
// gray image and cv2.equalizeHist's table pass, server.py:246 -- reads the batch once: 1 + 2 bytes per pixel instead of 2 + 2.
// Every source row of a wave's segment passes through its window registers exactly once (plus the segment's two halo rows,
// which belong to the neighbouring segments and are skipped): its 16 bytes go through the table (per-wave copy in LDS) there.
__device__ __forceinline__ uint32_t conv_lut4(const uint8_t* sl, uint32_t x)
{
    return (uint32_t)sl[x & 255u] | ((uint32_t)sl[(x >> 8) & 255u] << 8) |
           ((uint32_t)sl[(x >> 16) & 255u] << 16) | ((uint32_t)sl[x >> 24] << 24);
}

template <int OP>
__global__ __launch_bounds__(64 * kMaxWavesPerBlock) void conv3_lut_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                           uint8_t* __restrict__ dst2, ConvGeom g,
                                                                           const uint8_t* __restrict__ lut_dev, int64_t lut_stride)
{
    constexpr int C = 1;
    __shared__ uint32_t s_lut[kMaxWavesPerBlock][64];
    if (blockIdx.x >= g.main_blocks) {                       // ---- border bytes, one per thread (rows shorter than 44 bytes)
        unsigned img; int y, b;
        if (!border_item(g, blockIdx.x - g.main_blocks, img, y, b)) return;
        const uint8_t* simg = src + (int64_t)img * g.s_img;
        const int xl = reflect101(b - 1, g.w), xr = reflect101(b + 1, g.w);
        const uint8_t* rc = row_ptr(g, simg, y, 1);
        dst[(int64_t)img * g.d_img + (int64_t)y * g.d_row + b] = (uint8_t)conv3_byte<OP>(
            row_ptr(g, simg, y - 1, 1), rc, row_ptr(g, simg, y + 1, 1), xl, b, xr);
        dst2[(int64_t)img * g.d_img + (int64_t)y * g.d_row + b] = lut_dev[(int64_t)img * lut_stride + rc[b]];
        return;
    }
    unsigned img; int y0, y1, lane0;
    if (!wave_item(g, img, y0, y1, lane0)) return;
    const int lane = (int)(threadIdx.x & 63u), wave = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {
        uint32_t v;
        __builtin_memcpy(&v, lut_dev + (int64_t)img * lut_stride + 4 * lane, 4);
        s_lut[wave][lane] = v;                               // this wave's image's table; only this wave reads it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const uint8_t* sl = reinterpret_cast<const uint8_t*>(s_lut[wave]);
    const int gl = lane0 + lane;                             // 16-byte lane of the row
    if (gl >= g.nlanes) return;
    const int b0 = lane_b0(g, gl);
    const bool first = b0 == 0, last = b0 + 16 == g.L;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    uint8_t* d2img = dst2 + (int64_t)img * g.d_img;
    typedef Win<C, 1> W1;
    typedef Conv3<OP, C> Op;
    constexpr int U = kConv3U;                               // rows per step; loads run one step ahead
    const int nrows = y1 - y0;
    auto row = [&](int t) {                                  // t-th source row of the walk, t = 0 .. nrows + 1
        return row_ptr(g, simg, y0 + (t <= nrows + 1 ? t : nrows + 1) - 1, 1);
    };
    auto table_pass = [&](const W1& wv, int y) {             // the window's own 16 bytes (image row y) through the table
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = conv_lut4(sl, wv.w[W1::HD + k]);
        st16_nt(d2img + (int64_t)y * g.d_row + b0, o);
    };
    typename Op::Feat fa, fb;
    {
        W1 wa, wb;
        load_win<C, 1>(wa, row(0), b0, g.L);
        load_win<C, 1>(wb, row(1), b0, g.L);
        table_pass(wb, y0);
        fix_edges<C, 1>(wa, first, last);
        fix_edges<C, 1>(wb, first, last);
        Op::feat(fa, wa);
        Op::feat(fb, wb);
    }
    W1 nx[U];
#pragma unroll
    for (int j = 0; j < U; ++j) load_win<C, 1>(nx[j], row(2 + j), b0, g.L);
    for (int i = 0; i < nrows; i += U) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            typename Op::Feat fc;
            if (i + j + 1 < nrows) table_pass(nx[j], y0 + i + j + 1);   // nx[j] holds image row y0 + i + j + 1
            fix_edges<C, 1>(nx[j], first, last);
            Op::feat(fc, nx[j]);
            load_win<C, 1>(nx[j], row(i + U + 2 + j), b0, g.L);    // the slot is free: refill it with the row U steps further on
            if (i + j < nrows) st16_nt(dimg + (int64_t)(y0 + i + j) * g.d_row + b0, Op::out(fa, fb, fc));
            fa = fb;
            fb = fc;
        }
    }
}

// ---- separable fixed-point filter, K <= 7, register sliding window ---------------------------------------
// Column pass on ROW PAIRS: the Q8.8 row results of source rows (2p, 2p+1) of the walk are re-packed (2 v_perm per
// slot) into registers that hold the SAME column of both rows, so that one v_dot2_u32_u16 applies two taps.  The two
// output rows 2j and 2j+1 need the same (K+1)/2 pairs, with the taps shifted by one row:
//   even:  (ky0,ky1) (ky2,ky3) ... (ky[K-1],0)          odd:  (0,ky0) (ky1,ky2) ... (ky[K-2],ky[K-1])
// i.e. (K+1)/2 instructions per output byte instead of K multiply-adds.
template <int K> struct Taps { uint32_t kx[K]; uint32_t ce[(K + 1) / 2]; uint32_t co[(K + 1) / 2]; };

template <int K>
__device__ __forceinline__ uint32_t tap_y(const Taps<K>& t, int i)        // ky[i], 0 <= i < K
{
    return (i & 1) ? t.ce[i >> 1] >> 16 : t.ce[i >> 1] & 0xffffu;
}

template <int C, int K>
__device__ __forceinline__ void hpass(us2 (&h)[8], const Win<C, K / 2>& W, const Taps<K>& t)
{
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int p = slot_p(s);
        us2 acc = fld<C, K / 2>(W, p - C * (K / 2)) * (unsigned short)t.kx[0];
#pragma unroll
        for (int j = 1; j < K; ++j) acc += fld<C, K / 2>(W, p + C * (j - K / 2)) * (unsigned short)t.kx[j];
        h[s] = acc;                                           // Q8.8, <= 255*256: no 16-bit overflow
    }
}

// row results of two consecutive rows -> per-column pairs: pl[s] = (a[s].lo, b[s].lo), ph[s] = (a[s].hi, b[s].hi)
template <int C, int K>
__device__ __forceinline__ void make_pair(uint32_t (&pl)[8], uint32_t (&ph)[8], const Win<C, K / 2>& wa,
                                          const Win<C, K / 2>& wb, const Taps<K>& t)
{
    us2 ha[8], hb[8];
    hpass<C, K>(ha, wa, t);
    hpass<C, K>(hb, wb, t);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const uint32_t a = __builtin_bit_cast(uint32_t, ha[s]), b = __builtin_bit_cast(uint32_t, hb[s]);
        pl[s] = __builtin_amdgcn_perm(b, a, 0x05040100u);
        ph[s] = __builtin_amdgcn_perm(b, a, 0x07060302u);
    }
}

__device__ __forceinline__ uint32_t dot2(uint32_t pair, uint32_t taps, uint32_t acc)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, pair), __builtin_bit_cast(us2, taps), acc, false);
}

template <int C, int K>
__global__ __launch_bounds__(K <= 5 ? 64 * kMaxWavesPerBlock : 512) void sep_small_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                             ConvGeom g, Taps<K> t)
{
    constexpr int R = K / 2, NP = (K + 1) / 2;
#ifdef IPX_DEV
    const unsigned trace_id = 2u * (blockIdx.x * (unsigned)g.wpb + (threadIdx.x >> 6));
#endif
    if (blockIdx.x >= g.main_blocks) {                       // ---- border bytes, one per thread
        unsigned img; int y, b;
#ifdef IPX_DEV
        if (g.trace && (threadIdx.x & 63u) == 0) g.trace[trace_id] = __builtin_amdgcn_s_memrealtime();
#endif
        if (!border_item(g, blockIdx.x - g.main_blocks, img, y, b)) return;
        const uint8_t* simg = src + (int64_t)img * g.s_img;
        const int px = b / C, ch = b - px * C;
        int xo[K];
#pragma unroll
        for (int j = 0; j < K; ++j) xo[j] = reflect101(px + j - R, g.w) * C + ch;
        uint32_t acc = 32768u;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const uint8_t* rp = row_ptr(g, simg, y + i - R, R);
            uint32_t hsum = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) hsum += t.kx[j] * rp[xo[j]];
            acc += tap_y<K>(t, i) * hsum;
        }
        dst[(int64_t)img * g.d_img + (int64_t)y * g.d_row + b] = (uint8_t)(acc >> 16);
#ifdef IPX_DEV
        if (g.trace && (threadIdx.x & 63u) == 0) g.trace[trace_id + 1] = __builtin_amdgcn_s_memrealtime();
#endif
        return;
    }
    unsigned img; int y0, y1, lane0;
    if (!wave_item(g, img, y0, y1, lane0)) return;
    const int gl = lane0 + (int)(threadIdx.x & 63u);         // 16-byte lane of the row
    if (gl >= g.nlanes) return;
    const int b0 = lane_b0(g, gl);
    const bool first = b0 == 0, last = b0 + 16 == g.L;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    typedef Win<C, R> WR;
#ifdef IPX_DEV
    const bool tracer = g.trace && (threadIdx.x & 63u) == (gl == 1 ? 1u : 0u);
    if (tracer) g.trace[trace_id] = __builtin_amdgcn_s_memrealtime();
#endif

    const int nrows = y1 - y0;
    const int ys = y0;
    const int tlast = nrows - 1 + 2 * R;                     // last source row of the walk anybody needs
    auto row = [&](int tt) { return row_ptr(g, simg, ys + ((tt <= tlast ? tt : tlast) - R), R); };   // walk row tt

    uint32_t pl[NP][8], ph[NP][8];                           // ring of NP row pairs (statically indexed)
#pragma unroll
    for (int p = 0; p < NP - 1; ++p) {                       // prologue: pairs 0 .. NP-2
        WR wa, wb;
        load_win<C, R>(wa, row(2 * p), b0, g.L);
        load_win<C, R>(wb, row(2 * p + 1), b0, g.L);
        fix_edges<C, R>(wa, first, last);
        fix_edges<C, R>(wb, first, last);
        make_pair<C, K>(pl[p], ph[p], wa, wb, t);
    }
    constexpr int DP = 1;                                     // loads run DP row pairs ahead (DP divides NP: static slots).
                                                             // Measured: 3 pairs ahead cost 40 VGPRs (4 instead of 5 waves per SIMD)
                                                             // and were 0-3 % slower for K = 3, 5, 7
    WR nx[2 * DP];
#pragma unroll
    for (int s = 0; s < 2 * DP; ++s) load_win<C, R>(nx[s], row(2 * (NP - 1) + s), b0, g.L);
    for (int j0 = 0; 2 * j0 < nrows; j0 += NP) {
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            fix_edges<C, R>(nx[2 * (s % DP)], first, last);
            fix_edges<C, R>(nx[2 * (s % DP) + 1], first, last);
            make_pair<C, K>(pl[(s + NP - 1) % NP], ph[(s + NP - 1) % NP], nx[2 * (s % DP)], nx[2 * (s % DP) + 1], t);
            const int tn = 2 * (j0 + s + NP - 1 + DP);        // refill the two slots just consumed
            load_win<C, R>(nx[2 * (s % DP)], row(tn), b0, g.L);
            load_win<C, R>(nx[2 * (s % DP) + 1], row(tn + 1), b0, g.L);
            const int y = 2 * (j0 + s);
#pragma unroll
            for (int odd = 0; odd < 2; ++odd) {
                if (y + odd < nrows) {
                    u32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        uint32_t lo[2], hi[2];
#pragma unroll
                        for (int par = 0; par < 2; ++par) {
                            uint32_t al = 32768u, ah = 32768u;
#pragma unroll
                            for (int tt = 0; tt < NP; ++tt) {
                                const uint32_t c = odd ? t.co[tt] : t.ce[tt];
                                al = dot2(pl[(s + tt) % NP][2 * k + par], c, al);
                                ah = dot2(ph[(s + tt) % NP][2 * k + par], c, ah);
                            }
                            lo[par] = al;                     // taps sum to 256 -> (acc >> 16) <= 255
                            hi[par] = ah;
                        }
                        o[k] = pack_acc(lo[0], lo[1], hi[0], hi[1]);
                    }
                    st16_nt(dimg + (int64_t)(ys + y + odd) * g.d_row + b0, o);
                }
            }
        }
    }
#ifdef IPX_DEV
    if (tracer) g.trace[trace_id + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---- 3- and 5-tap filters whose taps are a power of two times small SYMMETRIC integers -----------------------------
// cv2.GaussianBlur's own 3 x 3 and 5 x 5 kernels are 64 * [1 2 1] and 16 * [1 4 6 4 1] in Q0.8.  With kx = gx * kx',
// ky = gy * ky', gx * gy = 2^m and 255 * sum(kx') * sum(ky') + rounding < 2^16, the whole filter is exact in PACKED
// 16-bit arithmetic: row pass B = sum kx'[j] s[x + j] and column pass S = sum ky'[i] B[y + i] both fit 16 bits, and
// (256-scaled sum + 2^15) >> 16 == (S + 2^(15 - m)) >> (16 - m).  Symmetry (outer tap 1) halves the multiplies:
// (a + e) + k1 (b + d) + k2 c is two packed adds and two packed multiply-adds for two pixels, in both passes -- about
// 100 vector instructions per 16 bytes and row against 150 for the general kernel (v_pk_mad_u16 row pass, v_dot2 column
// pass on row pairs).  On gray frames that is 10 % faster (75 against 83 us per 24 x 4K); on BGR frames it is not -- the walker's
// access pattern, not its instruction count, bounds it there (profiles/r02_notes.md).
struct Pow2Taps { uint32_t kx1, kx2, ky1, ky2, rnd, sh; };       // inner taps (K = 3: k1 = centre, k2 unused), rounding, shift

// Row pass and column pass of the packed filter, shared by the strip walker and the band-tile kernel.
template <int C, int K>
__device__ __forceinline__ void pow2_rowpass(us2 (&h)[8], const Win<C, K / 2>& W, unsigned short kx1, unsigned short kx2)
{
    constexpr int R = K / 2;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int p = slot_p(s);
        if (K == 3) {
            h[s] = (fld<C, R>(W, p - C) + fld<C, R>(W, p + C)) + fld<C, R>(W, p) * kx1;
        } else {
            const us2 o = fld<C, R>(W, p - 2 * C) + fld<C, R>(W, p + 2 * C);
            const us2 i = fld<C, R>(W, p - C) + fld<C, R>(W, p + C);
            h[s] = (o + i * kx1) + fld<C, R>(W, p) * kx2;
        }
    }
}

// hb = ring of the last K row-pass results, `slot` the newest (source row y + R of output row y): the 16 output bytes.
// No 16-bit field of the sum overflows (pow2_symmetric_taps checks 255 * sum kx' * sum ky' + rounding < 2^16), so the last two
// addends -- the outer row and the rounding constant -- go in with ONE 32-bit three-operand add on the packed pair (no carry
// leaves the low field).  HI8: the shift is 8 (cv2's 5-tap kernel, 16 * [1 4 6 4 1] on both axes), the result byte is the HIGH
// byte of every field: no shifts, one v_perm_b32 per output dword picks the four bytes -- 16 instructions less per 16 bytes and
// row than add, add, shift, shift-or (the band kernels issue vector instructions 70 % of their time: DESIGN.md section 4).
template <int K, bool HI8 = false>
__device__ __forceinline__ u32x4 pow2_colpass(const us2 (&hb)[K][8], int slot, unsigned short ky1, unsigned short ky2, us2 rnd, us2 sh)
{
    uint32_t o[8];
    // (slot is a compile-time value at every call site -- the callers' loops are unrolled by K -- and so are the ring indices)
    const int s1 = (slot + 1) % K, s2 = (slot + 2) % K, s3 = (slot + 3) % K, s4 = (slot + 4) % K;
    const uint32_t rnd32 = __builtin_bit_cast(uint32_t, rnd);
#pragma unroll
    for (int s = 0; s < 8; ++s) {                                // source rows y - R .. y + R = ring slots (slot + 1 + i) % K
        us2 m;
        if (K == 3) {
            m = hb[s2][s] * ky1 + hb[s1][s];
        } else {
            const us2 ii = hb[s2][s] + hb[s4][s];
            m = hb[s3][s] * ky2 + (ii * ky1 + hb[s1][s]);
        }
        o[s] = __builtin_bit_cast(uint32_t, m) + __builtin_bit_cast(uint32_t, hb[slot][s]) + rnd32;      // v_add3_u32
    }
    u32x4 ov;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (HI8)
            ov[k] = __builtin_amdgcn_perm(o[2 * k + 1], o[2 * k], 0x07030501u);     // bytes: even.lo >> 8, odd.lo >> 8, even.hi >> 8, odd.hi >> 8
        else
            ov[k] = pack_fields(__builtin_bit_cast(us2, o[2 * k]) >> sh, __builtin_bit_cast(us2, o[2 * k + 1]) >> sh);
    }
    return ov;
}

template <int C, int K, bool HI8 = false>
__global__ __launch_bounds__(64 * kMaxWavesPerBlock) void sep_pow2_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                       ConvGeom g, Pow2Taps t)
{
    constexpr int R = K / 2;
    typedef Win<C, R> WR;
    unsigned img; int y0, y1, lane0;
    if (!wave_item(g, img, y0, y1, lane0)) return;
    const int gl = lane0 + (int)(threadIdx.x & 63u);
    if (gl >= g.nlanes) return;
    const int b0 = lane_b0(g, gl);
    const bool first = b0 == 0, last = b0 + 16 == g.L;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    const int nrows = y1 - y0;
    const int tlast = nrows - 1 + 2 * R;                         // last source row of the walk anybody needs
    auto row = [&](int tt) { return row_ptr(g, simg, y0 + ((tt <= tlast ? tt : tlast) - R), R); };   // walk row tt
    const unsigned short kx1 = (unsigned short)t.kx1, kx2 = (unsigned short)t.kx2;
    const unsigned short ky1 = (unsigned short)t.ky1, ky2 = (unsigned short)t.ky2;
    const us2 rnd = {(unsigned short)t.rnd, (unsigned short)t.rnd};
    const us2 sh = {(unsigned short)t.sh, (unsigned short)t.sh};

    us2 hb[K][8];                                                // ring: row-pass results of the last K source rows
    auto hpass = [&](us2 (&h)[8], WR& W) {
        fix_edges<C, R>(W, first, last);
        pow2_rowpass<C, K>(h, W, kx1, kx2);
    };
    WR nx[K];
#pragma unroll
    for (int j = 0; j < K; ++j) load_win<C, R>(nx[j], row(j), b0, g.L);
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {                            // prologue: source rows 0 .. K - 2 -> ring slots 0 .. K - 2
        hpass(hb[j], nx[j]);
        load_win<C, R>(nx[j], row(K + j), b0, g.L);
    }
    // walk step tt (source row tt, tt >= K - 1) completes output row tt - (K - 1); unrolled by K: ring and window slot
    // tt % K are static
    for (int t0 = K - 1; t0 - (K - 1) < nrows; t0 += K) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const int tt = t0 + q;
            const int slot = (K - 1 + q) % K;                    // = tt % K
            hpass(hb[slot], nx[slot]);
            load_win<C, R>(nx[slot], row(tt + K), b0, g.L);
            const int yo = tt - (K - 1);
            if (yo < nrows) {
                const u32x4 ov = pow2_colpass<K, HI8>(hb, slot, ky1, ky2, rnd, sh);
                st16_nt(dimg + (int64_t)(y0 + yo) * g.d_row + b0, ov);
            }
        }
    }
}

// ---- the same packed filter on BAND tiles staged in LDS --------------------------------------------------------------
// Long-lived waves that each stream their own rows reach 5.2-5.6 TB/s on this part, short-lived blocks that load a
// contiguous band up front 5.9 (scripts/dev/bandbench.hip; a flat copy 6.2).  A block takes band_rows output rows x 4 KiB
// of row bytes: its (band_rows + 2R) x (4 KiB + 64) source bytes go straight from global memory into LDS (LDS-DMA, 1 KiB
// per wave instruction, nothing staged in registers), one barrier, then every lane walks its 16-byte column of the tile
// with the register ring of sep_pow2_kernel, reading rows from LDS.  Rows must be a multiple of 16 bytes (every lane
// aligned in LDS); other shapes take the strip walker.
struct BandGeom { ConvGeom g; int band_rows, nbands, ntx, tile_lanes, pitch; unsigned nblocks; };   // pitch = 16 tile_lanes + 64
constexpr int kBandLanes = 256;                            // lanes per tile at most (block = tile_lanes rounded up to whole waves)
constexpr int kMaxBandRows = 10;                          // output rows per tile at most (8 as a rule; make_band_geom)

struct BandTile {                                          // what a block of a band kernel works on (all wave-uniform)
    unsigned img;
    int y0, y1, rows_in;                                   // output rows [y0, y1), rows_in = y1 - y0 + 2R tile rows
    int lane0, xs;                                         // first 16-byte lane, first row byte held by the tile
};

// Decode the block's tile and stage its source bytes in LDS (LDS-DMA, 1 KiB per wave instruction); returns after the barrier.
template <int R>
__device__ __forceinline__ BandTile band_stage(const BandGeom& bg, const uint8_t* __restrict__ src, uint8_t* s_tile)
{
    const ConvGeom& g = bg.g;
    BandTile t;
    const unsigned blk = xcd_remap(blockIdx.x, bg.nblocks);
    const int xt = (int)(blk % (unsigned)bg.ntx);
    const unsigned rest = blk / (unsigned)bg.ntx;
    const int band = (int)(rest % (unsigned)bg.nbands);
    t.img = rest / (unsigned)bg.nbands;
    t.y0 = band * bg.band_rows;
    t.y1 = t.y0 + bg.band_rows < g.h ? t.y0 + bg.band_rows : g.h;
    t.rows_in = t.y1 - t.y0 + 2 * R;
    t.lane0 = xt * bg.tile_lanes;
    t.xs = 16 * t.lane0 - 32 > 0 ? 16 * t.lane0 - 32 : 0;                    // tile = row bytes [xs, xe), both multiples of 16
    const int xe = 16 * (t.lane0 + bg.tile_lanes) + 32 < g.L ? 16 * (t.lane0 + bg.tile_lanes) + 32 : g.L;
    const int nch = (xe - t.xs) >> 4;                                        // 16-byte chunks per tile row (<= tile_lanes + 4)
    const uint8_t* simg = src + (int64_t)t.img * g.s_img;
    const int wave = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    const int nwaves = (int)(blockDim.x >> 6);
    for (int r = wave; r < t.rows_in; r += nwaves) {                         // wave w stages tile rows w, w + #waves, ...
        const uint8_t* rp = row_ptr(g, simg, t.y0 - R + r, R) + t.xs;
        for (int c0 = 0; c0 < nch; c0 += 64) {
            if (c0 + lane < nch)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(rp + 16 * (c0 + lane)),
                                                 (void __attribute__((address_space(3)))*)(s_tile + r * bg.pitch + 16 * c0), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    return t;
}

// A lane's window of tile row r: its 16 bytes at `off`, halo dwords at `lo` / `hi` (redirected inside the row at its ends)
template <int C, int R>
__device__ __forceinline__ void band_win(Win<C, R>& W, const uint8_t* s_tile, int pitch, int r, int off, int lo, int hi)
{
    constexpr int HD = Win<C, R>::HD;
    const uint8_t* tr = s_tile + r * pitch;
    const u32x4 c = *(const u32x4*)(tr + off);
    __builtin_memcpy(&W.w[HD], &c, 16);
#pragma unroll
    for (int i = 0; i < HD; ++i) {
        W.w[i] = *(const uint32_t*)(tr + lo + 4 * i);
        W.w[HD + 4 + i] = *(const uint32_t*)(tr + hi + 4 * i);
    }
}

template <int C, int K, bool HI8 = false>
__global__ __launch_bounds__(kBandLanes) void sep_pow2_band_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                   BandGeom bg, Pow2Taps t)
{
    constexpr int R = K / 2;
    typedef Win<C, R> WR;
    constexpr int HD = WR::HD;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];
    const ConvGeom& g = bg.g;
    const BandTile bt = band_stage<R>(bg, src, s_tile);
    const int gl = bt.lane0 + (int)threadIdx.x;
    if ((int)threadIdx.x >= bg.tile_lanes || gl >= g.nlanes) return;
    const int b0 = 16 * gl;
    const bool first = b0 == 0, last = b0 + 16 == g.L;
    const int off = b0 - bt.xs;
    const int lo = (b0 >= 4 * HD ? b0 - 4 * HD : b0) - bt.xs;
    const int hi = (b0 + 16 + 4 * HD <= g.L ? b0 + 16 : b0 + 16 - 4 * HD) - bt.xs;
    uint8_t* dimg = dst + (int64_t)bt.img * g.d_img;
    const unsigned short kx1 = (unsigned short)t.kx1, kx2 = (unsigned short)t.kx2;
    const unsigned short ky1 = (unsigned short)t.ky1, ky2 = (unsigned short)t.ky2;
    const us2 rnd = {(unsigned short)t.rnd, (unsigned short)t.rnd};
    const us2 sh = {(unsigned short)t.sh, (unsigned short)t.sh};

    us2 hb[K][8];
    auto hrow = [&](us2 (&h)[8], int r) {                                    // row pass of tile row r
        WR W;
        band_win<C, R>(W, s_tile, bg.pitch, r, off, lo, hi);
        fix_edges<C, R>(W, first, last);
        pow2_rowpass<C, K>(h, W, kx1, kx2);
    };
#pragma unroll
    for (int j = 0; j < K - 1; ++j) hrow(hb[j], j);
    const int nrows = bt.y1 - bt.y0;
    for (int t0 = K - 1; t0 - (K - 1) < nrows; t0 += K) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const int tt = t0 + q;
            const int slot = (K - 1 + q) % K;                                // = tt % K
            const int yo = tt - (K - 1);
            if (yo < nrows) {
                hrow(hb[slot], tt);
                const u32x4 ov = pow2_colpass<K, HI8>(hb, slot, ky1, ky2, rnd, sh);
                st16_nt(dimg + (int64_t)(bt.y0 + yo) * g.d_row + b0, ov);
            }
        }
    }
}

// The packed Gaussian on BGR band tiles with a SECOND output from the same tile: the gray image, cv2.cvtColor(BGR2GRAY).
// BASELINE.json configs[1] blurs the colour frame and converts it to gray (for Sobel and equalizeHist): both read the whole
// colour batch, 30 % of the step's bytes.  The tile is in LDS anyway: after the filter walk (before it: 294 against 290 us per
// 24 x 4K), the block converts the pixels of
// its 8 output rows whose first byte lies in the tile's own byte range -- groups of 8 pixels = 24 tile bytes (8-byte aligned in
// LDS; a group reaches at most 23 bytes into the 32-byte halo), three ds_read_b64, a pixel's three bytes as one dword by
// v_alignbyte_b32 and the Q15 sum (weights doubled: the result is byte 2 of the sum) as two byte dot products (7470 = 29 * 256 + 46, 38470 = 150 * 256 + 70, 19596 = 76 * 256 + 140:
// exact), one 8-byte store per group.  A lane takes one group column through the tile's rows (dealing the (group, row) items
// over all lanes of the block instead measured slower: 306 against 294 us per 24 x 4K).
struct GrayOut { uint8_t* p; int64_t row, img; };

// The gray phase of a BGR band tile (see sep_pow2_band_gray_kernel): cv2.cvtColor(BGR2GRAY) of the pixels of the tile's output rows
// whose first byte lies in the tile's own byte range, from the tile in LDS; called by every thread of the block.
template <int R>
__device__ __forceinline__ void band_gray_phase(const BandGeom& bg, const BandTile& bt, const GrayOut& gr, const uint8_t* s_tile)
{
    const ConvGeom& g = bg.g;
    const int nrows = bt.y1 - bt.y0;
    const int own0 = 16 * bt.lane0;
    const int own1 = 16 * (bt.lane0 + bg.tile_lanes) < g.L ? 16 * (bt.lane0 + bg.tile_lanes) : g.L;
    const int j0 = (own0 + 23) / 24, j1 = (own1 + 23) / 24;               // groups whose first byte 24 j lies in [own0, own1)
    const int jj = j0 + (int)threadIdx.x;
    if (jj < j1) {
        const int px0 = 8 * jj;
        const uint8_t* tp = s_tile + R * bg.pitch + (24 * jj - bt.xs);
        uint8_t* gp = gr.p + (int64_t)bt.img * gr.img + (int64_t)bt.y0 * gr.row + px0;
        // Q15 weights DOUBLED (7470 = 29 * 256 + 46, 38470 = 150 * 256 + 70, 19596 = 76 * 256 + 140; rounding 32768): the same
        // quotient, now (sum >> 16) = byte 2 of the sum, which v_perm_b32 picks while it packs -- no shift per pixel
        const uint32_t clo = 46u | (70u << 8) | (140u << 16), chi = 29u | (150u << 8) | (76u << 16);   // B, G, R
#pragma unroll
        for (int r = 0; r < kMaxBandRows; ++r) {                          // (band tiles are 8 output rows, 9 or 10 for a lone frame; unrolled: the rows' LDS reads overlap)
            if (r < nrows) {                                                 // (no break: the loop unrolls completely)
                const u32x2 q0 = *(const u32x2*)(tp), q1 = *(const u32x2*)(tp + 8), q2 = *(const u32x2*)(tp + 16);
                const uint32_t d[6] = {q0[0], q0[1], q1[0], q1[1], q2[0], q2[1]};
                uint32_t y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = (3 * i) >> 2, sh = (3 * i) & 3;
                    const uint32_t px = sh == 0 ? d[k] : __builtin_amdgcn_alignbyte(d[k + 1 < 6 ? k + 1 : 5], d[k], (uint32_t)sh);
                    const uint32_t lo = __builtin_amdgcn_udot4(px, clo, 32768u, false);
                    const uint32_t hi = __builtin_amdgcn_udot4(px, chi, 0u, false);
                    y[i] = (hi << 8) + lo;                                    // gray value in bits 16..23
                }
                const uint32_t p01 = __builtin_amdgcn_perm(y[1], y[0], 0x0c0c0602u), p23 = __builtin_amdgcn_perm(y[3], y[2], 0x06020c0cu);
                const uint32_t p45 = __builtin_amdgcn_perm(y[5], y[4], 0x0c0c0602u), p67 = __builtin_amdgcn_perm(y[7], y[6], 0x06020c0cu);
                const u32x2 o = {p01 | p23, p45 | p67};
                if (px0 + 8 <= g.w) {
                    __builtin_nontemporal_store(o, (u32x2_u*)gp);
                } else {
                    for (int i = 0; px0 + i < g.w; ++i) gp[i] = (uint8_t)(y[i] >> 16);
                }
                tp += bg.pitch;
                gp += gr.row;
            }
        }
    }
}


template <int K, bool HI8 = false>
__global__ __launch_bounds__(kBandLanes) void sep_pow2_band_gray_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                        BandGeom bg, Pow2Taps t, GrayOut gr)
{
    constexpr int C = 3, R = K / 2;
    typedef Win<C, R> WR;
    constexpr int HD = WR::HD;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];
    const ConvGeom& g = bg.g;
    const BandTile bt = band_stage<R>(bg, src, s_tile);
    const int nrows = bt.y1 - bt.y0;
    const int gl = bt.lane0 + (int)threadIdx.x;
    if (!((int)threadIdx.x >= bg.tile_lanes || gl >= g.nlanes)) {
    const int b0 = 16 * gl;
    const bool first = b0 == 0, last = b0 + 16 == g.L;
    const int off = b0 - bt.xs;
    const int lo = (b0 >= 4 * HD ? b0 - 4 * HD : b0) - bt.xs;
    const int hi = (b0 + 16 + 4 * HD <= g.L ? b0 + 16 : b0 + 16 - 4 * HD) - bt.xs;
    uint8_t* dimg = dst + (int64_t)bt.img * g.d_img;
    const unsigned short kx1 = (unsigned short)t.kx1, kx2 = (unsigned short)t.kx2;
    const unsigned short ky1 = (unsigned short)t.ky1, ky2 = (unsigned short)t.ky2;
    const us2 rnd = {(unsigned short)t.rnd, (unsigned short)t.rnd};
    const us2 sh = {(unsigned short)t.sh, (unsigned short)t.sh};
    us2 hb[K][8];
    auto hrow = [&](us2 (&h)[8], int r) {                                    // row pass of tile row r
        WR W;
        band_win<C, R>(W, s_tile, bg.pitch, r, off, lo, hi);
        fix_edges<C, R>(W, first, last);
        pow2_rowpass<C, K>(h, W, kx1, kx2);
    };
#pragma unroll
    for (int j = 0; j < K - 1; ++j) hrow(hb[j], j);
    for (int t0 = K - 1; t0 - (K - 1) < nrows; t0 += K) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const int tt = t0 + q;
            const int slot = (K - 1 + q) % K;                                // = tt % K
            const int yo = tt - (K - 1);
            if (yo < nrows) {
                hrow(hb[slot], tt);
                const u32x4 ov = pow2_colpass<K, HI8>(hb, slot, ky1, ky2, rnd, sh);
                st16_nt(dimg + (int64_t)(bt.y0 + yo) * g.d_row + b0, ov);
            }
        }
    }
    }
    band_gray_phase<R>(bg, bt, gr, s_tile);
}

// THREE outputs from one BGR band tile: cv2's 5-tap Gaussian, the Laplacian sharpening of server.py:270-276 (clip(5c - n - s - e - w),
// Conv3<OP_SHARPEN>: the same row features and vertical combine as conv3_kernel's) and the gray image.  BASELINE.json configs[2] runs
// all three on every colour frame -- three reads of the colour batch (4 + 6 + 6 bytes per pixel) become one (3 + 3 + 3 + 1).  The walk
// over the tile's 12 rows is unrolled completely (band tiles are 8 output rows), so the ring of 5 Gaussian row results and the ring
// of 3 sharpening row features are both indexed statically; tile row t is the Gaussian's newest row for output row t - 4 and the
// sharpening's lower neighbour for output row t - 3.
struct SharpOut { uint8_t* p; int64_t row, img; };

template <bool HI8, int ROWS = 8>
__global__ __launch_bounds__(kBandLanes) void gauss5_sharpen_gray_band_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                              BandGeom bg, Pow2Taps t, GrayOut gr, SharpOut sp)
{
    constexpr int C = 3, K = 5, R = 2, kRows = ROWS + 2 * R;       // ROWS = the tile height make_band_geom chose (8 or 9)
    typedef Win<C, R> WR;
    typedef Conv3<OP_SHARPEN, C> Sh;
    constexpr int HD = WR::HD;
    static_assert(HD == 2 && Win<C, 1>::HD == 1, "window layouts this kernel converts between");
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];
    const ConvGeom& g = bg.g;
    const BandTile bt = band_stage<R>(bg, src, s_tile);
    const int gl = bt.lane0 + (int)threadIdx.x;
    if (!((int)threadIdx.x >= bg.tile_lanes || gl >= g.nlanes)) {
        const int b0 = 16 * gl;
        const bool first = b0 == 0, last = b0 + 16 == g.L;
        const int off = b0 - bt.xs;
        const int lo = (b0 >= 4 * HD ? b0 - 4 * HD : b0) - bt.xs;
        const int hi = (b0 + 16 + 4 * HD <= g.L ? b0 + 16 : b0 + 16 - 4 * HD) - bt.xs;
        uint8_t* dimg = dst + (int64_t)bt.img * g.d_img + (int64_t)bt.y0 * g.d_row + b0;
        uint8_t* simg = sp.p + (int64_t)bt.img * sp.img + (int64_t)bt.y0 * sp.row + b0;
        const unsigned short kx1 = (unsigned short)t.kx1, kx2 = (unsigned short)t.kx2;
        const unsigned short ky1 = (unsigned short)t.ky1, ky2 = (unsigned short)t.ky2;
        const us2 rnd = {(unsigned short)t.rnd, (unsigned short)t.rnd};
        const us2 sh = {(unsigned short)t.sh, (unsigned short)t.sh};
        us2 hb[K][8];
        typename Sh::Feat F[3];
#pragma unroll
        for (int tt = 0; tt < kRows; ++tt) {
            if (tt < bt.rows_in) {                                            // (a last band of fewer rows)
                WR W;
                band_win<C, R>(W, s_tile, bg.pitch, tt, off, lo, hi);
                fix_edges<C, R>(W, first, last);
                pow2_rowpass<C, K>(hb[tt % K], W, kx1, kx2);
                if (tt >= 1 && tt <= kRows - 2) {                             // rows that are a centre or a neighbour of the 3 x 3 operator
                    Win<C, 1> W1;
                    W1.w[0] = W.w[1];
#pragma unroll
                    for (int i = 0; i < 4; ++i) W1.w[1 + i] = W.w[2 + i];
                    W1.w[5] = W.w[6];
                    Sh::feat(F[tt % 3], W1);
                }
                if (tt >= K - 1) {
                    const u32x4 ov = pow2_colpass<K, HI8>(hb, tt % K, ky1, ky2, rnd, sh);
                    st16_nt(dimg + (int64_t)(tt - (K - 1)) * g.d_row, ov);
                }
                if (tt >= 3 && tt + 1 < bt.rows_in) {                         // sharpening of output row tt - 3: tile rows tt - 2, tt - 1, tt
                    const u32x4 so = Sh::out(F[(tt + 1) % 3], F[(tt + 2) % 3], F[tt % 3]);
                    st16_nt(simg + (int64_t)(tt - 3) * sp.row, so);
                }
            }
        }
    }
    band_gray_phase<R>(bg, bt, gr, s_tile);
}

// The 3 x 3 operators on the same band tiles (sharpen / box / Sobel on BGR frames: rows of >= 4 KiB).
template <int OP, int C>
__global__ __launch_bounds__(kBandLanes) void conv3_band_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, BandGeom bg)
{
    typedef Win<C, 1> W1;
    typedef Conv3<OP, C> Op;
    constexpr int HD = W1::HD;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];
    const ConvGeom& g = bg.g;
    const BandTile bt = band_stage<1>(bg, src, s_tile);
    const int gl = bt.lane0 + (int)threadIdx.x;
    if ((int)threadIdx.x >= bg.tile_lanes || gl >= g.nlanes) return;
    const int b0 = 16 * gl;
    const bool first = b0 == 0, last = b0 + 16 == g.L;
    const int off = b0 - bt.xs;
    const int lo = (b0 >= 4 * HD ? b0 - 4 * HD : b0) - bt.xs;
    const int hi = (b0 + 16 + 4 * HD <= g.L ? b0 + 16 : b0 + 16 - 4 * HD) - bt.xs;
    uint8_t* dimg = dst + (int64_t)bt.img * g.d_img;
    typename Op::Feat f[3];
    auto frow = [&](typename Op::Feat& ft, int r) {
        W1 W;
        band_win<C, 1>(W, s_tile, bg.pitch, r, off, lo, hi);
        fix_edges<C, 1>(W, first, last);
        Op::feat(ft, W);
    };
    frow(f[0], 0);
    frow(f[1], 1);
    const int nrows = bt.y1 - bt.y0;
    for (int t0 = 2; t0 - 2 < nrows; t0 += 3) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int tt = t0 + q, slot = (2 + q) % 3, yo = tt - 2;
            if (yo < nrows) {
                frow(f[slot], tt);
                st16_nt(dimg + (int64_t)(bt.y0 + yo) * g.d_row + b0, Op::out(f[(slot + 1) % 3], f[(slot + 2) % 3], f[slot]));
            }
        }
    }
}

// ---- erode / dilate with a small structuring element (cv2.erode / cv2.dilate, server.py:433-434) -------------------
// Same strip walker, gray images.  A structuring element of K x K (K = 3 or 5) whose non-empty rows use at most two
// distinct column masks (every rectangle, cross and OpenCV ellipse of that size does): per source row each lane forms,
// for its 8 field pairs, the horizontal min / max over mask A and over mask B (v_pk_min_u16 / v_pk_max_u16 on fields);
// the vertical combine picks A or B per structuring-element row from a register ring of the last K rows.  Pixels
// outside the image do not take part (OpenCV's default morphology border): their bytes and rows are the neutral value.
struct MorphSE { uint32_t mask[2]; int sel[5]; };            // sel[i]: 0 = row i of the element is empty, 1 = mask A, 2 = mask B

template <bool DILATE> __device__ __forceinline__ us2 mm(us2 a, us2 b)
{
    return DILATE ? __builtin_elementwise_max(a, b) : __builtin_elementwise_min(a, b);
}

// SHAPE names the two column masks at compile time where they are the usual ones (the generic form tests a mask bit per
// tap and slot: 27 vector instructions per pixel for 5 x 5, VALU-bound): 1 = mask A is the centre column and mask B the
// full row (OpenCV's 3 x 3 and 5 x 5 ellipses, crosses), 2 = mask A is the full row and there is no mask B (rectangles);
// the full-row min / max of a lane's slot pairs p, p + 1 share the taps p - R + 1 .. p + R.  0 = any two masks.
template <bool DILATE, int R, int SHAPE>
__global__ __launch_bounds__(256) void morph_small_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                          ConvGeom g, MorphSE se)
{
    constexpr int K = 2 * R + 1;
    typedef Win<1, R> WR;
    constexpr int HD = WR::HD;
    constexpr uint32_t NEU = DILATE ? 0u : 0xffffffffu;          // neutral bytes
    const us2 neu = __builtin_bit_cast(us2, DILATE ? 0u : 0x00ff00ffu);
    unsigned img; int y0, y1, lane0;
    if (!wave_item(g, img, y0, y1, lane0)) return;
    const int gl = lane0 + (int)(threadIdx.x & 63u);
    if (gl >= g.nlanes) return;
    const int b0 = lane_b0(g, gl);
    const bool first = b0 == 0, last = b0 + 16 == g.L;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    const int nrows = y1 - y0;
    const bool useB = SHAPE == 1 || (SHAPE == 0 && se.mask[1] != 0);   // wave-uniform

    auto row_y = [&](int t) { return y0 - R + t; };              // source row of walk step t
    auto row_ptr_cl = [&](int t) {
        int yy = row_y(t);
        yy = yy < 0 ? 0 : yy > g.h - 1 ? g.h - 1 : yy;           // rows outside the image are loaded clamped and then ignored
        return simg + (int64_t)yy * g.s_row;
    };
    us2 ha[K][8], hb[K][8];                                      // ring: horizontal results of the last K source rows
    auto feat = [&](us2 (&fa)[8], us2 (&fb)[8], WR& W, int t) {
        const int yy = row_y(t);
        if (yy < 0 || yy >= g.h) {                               // wave-uniform
#pragma unroll
            for (int s = 0; s < 8; ++s) { fa[s] = neu; fb[s] = neu; }
            return;
        }
        if (first) {
#pragma unroll
            for (int i = 0; i < HD; ++i) W.w[i] = NEU;
        }
        if (last) {
#pragma unroll
            for (int i = 0; i < HD; ++i) W.w[HD + 4 + i] = NEU;
        }
        if (SHAPE != 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = 4 * k;                             // slots 2k, 2k + 1 = fields p, p + 1
                us2 common = fld<1, R>(W, p - R + 1);
#pragma unroll
                for (int j = p - R + 2; j <= p + R; ++j) common = mm<DILATE>(common, fld<1, R>(W, j));
                const us2 full0 = mm<DILATE>(common, fld<1, R>(W, p - R)), full1 = mm<DILATE>(common, fld<1, R>(W, p + R + 1));
                if (SHAPE == 1) {
                    fa[2 * k] = fld<1, R>(W, p); fa[2 * k + 1] = fld<1, R>(W, p + 1);
                    fb[2 * k] = full0; fb[2 * k + 1] = full1;
                } else {
                    fa[2 * k] = full0; fa[2 * k + 1] = full1;
                    fb[2 * k] = neu; fb[2 * k + 1] = neu;
                }
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int p = slot_p(s);
            us2 a = neu, b = neu;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const us2 f = fld<1, R>(W, p + j - R);
                if (se.mask[0] >> j & 1u) a = mm<DILATE>(a, f);      // wave-uniform
                if (useB && (se.mask[1] >> j & 1u)) b = mm<DILATE>(b, f);
            }
            fa[s] = a;
            fb[s] = b;
        }
    };
    WR nx[K];
#pragma unroll
    for (int j = 0; j < K; ++j) load_win<1, R>(nx[j], row_ptr_cl(j), b0, g.L);
    // prologue: the first K - 1 source rows fill ring slots 0 .. K - 2 (and their windows are refilled)
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
        feat(ha[j], hb[j], nx[j], j);
        load_win<1, R>(nx[j], row_ptr_cl(K + j), b0, g.L);
    }
    // step t (source row t, t >= K - 1) completes output row t - (K - 1); the loop is unrolled by K so that ring and window
    // slots are static: step t uses slot t % K
    for (int t0 = K - 1; t0 - (K - 1) < nrows; t0 += K) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const int t = t0 + q;
            constexpr int dummy = 0; (void)dummy;
            const int slot = (K - 1 + q) % K;                    // = t % K because t0 = K - 1 (mod K)
            feat(ha[slot], hb[slot], nx[slot], t);
            load_win<1, R>(nx[slot], row_ptr_cl(t + K), b0, g.L);
            const int yo = t - (K - 1);
            if (yo < nrows) {
                us2 o[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) o[s] = neu;
#pragma unroll
                for (int i = 0; i < K; ++i) {                    // element row i meets source row yo + i = ring slot (slot + 1 + i) % K
                    const int rs = (slot + 1 + i) % K;
                    if (se.sel[i] == 1) {
#pragma unroll
                        for (int s = 0; s < 8; ++s) o[s] = mm<DILATE>(o[s], ha[rs][s]);
                    } else if (SHAPE != 2 && se.sel[i] == 2) {
#pragma unroll
                        for (int s = 0; s < 8; ++s) o[s] = mm<DILATE>(o[s], hb[rs][s]);
                    }
                }
                u32x4 ov;
#pragma unroll
                for (int k = 0; k < 4; ++k) ov[k] = pack_fields(o[2 * k], o[2 * k + 1]);
                st16_nt(dimg + (int64_t)(y0 + yo) * g.d_row + b0, ov);
            }
        }
    }
}

// ---- 3 x 3 median (cv2.medianBlur(gray, 3), server.py:436) on the strip walker --------------------------------------
// Gray images, BORDER_REPLICATE.  Per output row a lane sorts the 3 vertical values of each of its 16 columns' field pairs
// (lo, mid, hi: 6 packed min / max) and takes, per pixel, med3(max of the three neighbouring lo, med3 of the mid, min of
// the hi) -- the classic exact 3 x 3 median from sorted columns; neighbouring pixels share the pair min / max.  About 200
// vector instructions per 16 pixels (the tile kernel in ipx_rank.hip sorts 9 values per pixel from scratch).
__device__ __forceinline__ us2 umin2(us2 a, us2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ us2 umax2(us2 a, us2 b) { return __builtin_elementwise_max(a, b); }

__global__ __launch_bounds__(64 * kMaxWavesPerBlock) void median3_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, ConvGeom g)
{
    constexpr int K = 3, NQ = 16;                                // fields q = -1 .. 14
    typedef Win<1, 1> WR;
    unsigned img; int y0, y1, lane0;
    if (!wave_item(g, img, y0, y1, lane0)) return;
    const int gl = lane0 + (int)(threadIdx.x & 63u);
    if (gl >= g.nlanes) return;
    const int b0 = lane_b0(g, gl);
    const bool first = b0 == 0, last = b0 + 16 == g.L;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    const int nrows = y1 - y0;
    auto row = [&](int t) {                                      // walk step t = source row y0 - 1 + t, replicated at the ends
        int yy = y0 - 1 + t;
        yy = yy < 0 ? 0 : yy > g.h - 1 ? g.h - 1 : yy;
        return simg + (int64_t)yy * g.s_row;
    };
    us2 f[K][NQ];                                                // ring: the field pairs of the last 3 source rows
    auto fields = [&](us2 (&o)[NQ], WR& W) {
        if (first) W.w[0] = W.w[1] << 24;                        // byte -1 = byte 0
        if (last) W.w[5] = W.w[4] >> 24;                         // byte 16 = byte 15
#pragma unroll
        for (int q = 0; q < NQ; ++q) o[q] = fld<1, 1>(W, q - 1);
    };
    WR nx[K];
#pragma unroll
    for (int j = 0; j < K; ++j) load_win<1, 1>(nx[j], row(j), b0, g.L);
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
        fields(f[j], nx[j]);
        load_win<1, 1>(nx[j], row(K + j), b0, g.L);
    }
    for (int t0 = K - 1; t0 - (K - 1) < nrows; t0 += K) {
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int t = t0 + u;
            const int slot = (K - 1 + u) % K;                    // = t % K
            fields(f[slot], nx[slot]);
            load_win<1, 1>(nx[slot], row(t + K), b0, g.L);
            const int yo = t - (K - 1);
            if (yo < nrows) {
                us2 lo[NQ], mid[NQ], hi[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {                   // sort the column (the order of the three rows does not matter)
                    const us2 a = f[0][q], b = f[1][q], c = f[2][q];
                    const us2 mn = umin2(a, b), mx = umax2(a, b);
                    lo[q] = umin2(mn, c);
                    hi[q] = umax2(mx, c);
                    mid[q] = umax2(mn, umin2(mx, c));
                }
                us2 o[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {                    // slots 2k, 2k + 1 = fields p, p + 1 with p = 4k; array index = q + 1
                    const int i = 4 * k + 1;
                    const us2 ml = umax2(lo[i], lo[i + 1]), mh = umin2(hi[i], hi[i + 1]);
                    const us2 mn = umin2(mid[i], mid[i + 1]), mx = umax2(mid[i], mid[i + 1]);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int j = e ? i + 2 : i - 1;         // the third column: p - 1 for slot p, p + 2 for slot p + 1
                        const us2 A = umax2(ml, lo[j]), C = umin2(mh, hi[j]);
                        const us2 B = umax2(mn, umin2(mx, mid[j]));
                        o[2 * k + e] = umax2(umin2(A, B), umin2(umax2(A, B), C));
                    }
                }
                u32x4 ov;
#pragma unroll
                for (int k = 0; k < 4; ++k) ov[k] = pack_fields(o[2 * k], o[2 * k + 1]);
                st16_nt(dimg + (int64_t)(y0 + yo) * g.d_row + b0, ov);
            }
        }
    }
}

// ---- 5 x 5 median (cv2.medianBlur(gray, 5), server.py:436 with denoise_manual's default kernel) on the strip walker ----
// Gray images, BORDER_REPLICATE.  The min / max network comes from scripts/gen_median5_net.py (verified there and in
// tests/test_median5_net.py on all 2^25 0-1 inputs) and is built so that neighbouring windows share most of it:
//   A) every column's 5 vertical values are sorted once (a column belongs to 5 windows),
//   B) the sorted columns (2i, 2i + 1) are merged into sorted 10-lists once (such a pair belongs to 4 windows),
//   C) of two neighbouring pairs (four columns) the middle six of their 20 values are selected once: they belong to the
//      window that adds the column on their right (even x: pairs (x-2, x-1), (x, x+1) and column x+2) and to the one that
//      adds the column on their left (odd x: column x-2 and pairs (x-1, x), (x+1, x+2)),
//   D) per window, the median is rank 5 of those six and its fifth sorted column.
// On field pairs (byte p, byte p+2): E_d = bytes (4d, 4d+2) and O_d = bytes (4d+1, 4d+3) of dword d sort as two columns
// each, merge(E_d, O_d) = P_d holds pairs (4d, 4d+1) | (4d+2, 4d+3), and the lists a window needs that straddle two dwords
// are v_alignbit_b32 of neighbouring registers (Q_d = P_d.hi | P_d+1.lo; M_d = middle(P_d, Q_d) serves the odd windows
// of dword d as it is and the even ones as M_d-1.hi | M_d.lo).  Per 16 pixels: 12 column sorts, 6 merges, 5 middles,
// 8 finals = 216 + 156 + 180 + 80 packed min / max, 114 v_alignbit and ~60 moves and v_perm -- 50 per pixel, where the
// tile kernel in ipx_rank.hip spends 2 x 120 on a from-scratch selection per pixel pair.
#define M5_T us2
#define M5_MIN(a, b) __builtin_elementwise_min(a, b)
#define M5_MAX(a, b) __builtin_elementwise_max(a, b)
#define M5_FN __device__ __forceinline__
#include "ipx_median5_net.inc"
#undef M5_T
#undef M5_MIN
#undef M5_MAX
#undef M5_FN

__device__ __forceinline__ us2 hi_lo(us2 left, us2 right)        // (left's high half, right's low half): one v_alignbit_b32
{
    return __builtin_bit_cast(us2, __builtin_amdgcn_alignbit(__builtin_bit_cast(uint32_t, right), __builtin_bit_cast(uint32_t, left), 16));
}

template <int PF>
__global__ __launch_bounds__(256) void median5_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, ConvGeom g)
{
    constexpr int K = 5, ND = 6;                                 // dwords d = 0 .. 5 of the window: row bytes b0 - 4 .. b0 + 19
    constexpr int NR = 2 * ND;                                   // field pairs per row: E_d, O_d
    typedef Win<1, 2> WR;
    static_assert(WR::ND == ND, "one halo dword on either side");
    unsigned img; int y0, y1, lane0;
    if (!wave_item(g, img, y0, y1, lane0)) return;
    const int gl = lane0 + (int)(threadIdx.x & 63u);
    if (gl >= g.nlanes) return;
    const int b0 = lane_b0(g, gl);
    const bool first = b0 == 0, last = b0 + 16 == g.L;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    const int nrows = y1 - y0;
    auto row = [&](int t) {                                      // walk step t = source row y0 - 2 + t, replicated at the ends
        int yy = y0 - 2 + t;
        yy = yy < 0 ? 0 : yy > g.h - 1 ? g.h - 1 : yy;
        return simg + (int64_t)yy * g.s_row;
    };
    auto even = [](uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c020c00u); };   // bytes (0, 2) as 2 x u16
    auto odd = [](uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c030c01u); };    // bytes (1, 3)
    uint32_t ring[K][NR];                                        // the field pairs of the last 5 source rows (raw dwords in the
                                                                 // ring and fields per step: 149 instead of 130 VGPRs and 1 % slower)
    auto settle = [&](uint32_t (&o)[NR], WR& W) {
        if (first) W.w[0] = (W.w[1] & 255u) * 0x01010101u;       // bytes -4 .. -1 = byte 0
        if (last) W.w[5] = (W.w[4] >> 24) * 0x01010101u;         // bytes 16 .. 19 = byte 15
#pragma unroll
        for (int d = 0; d < ND; ++d) { o[2 * d] = even(W.w[d]); o[2 * d + 1] = odd(W.w[d]); }
    };
    {
        WR w0[K - 1];
#pragma unroll
        for (int j = 0; j < K - 1; ++j) load_win<1, 2>(w0[j], row(j), b0, g.L);
#pragma unroll
        for (int j = 0; j < K - 1; ++j) settle(ring[j + 1], w0[j]);
    }
    WR nx[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) load_win<1, 2>(nx[j], row(K - 1 + j), b0, g.L);
    for (int t0 = K - 1; t0 - (K - 1) < nrows; t0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t = t0 + u, yo = t - (K - 1);
            if (yo >= nrows) break;                              // wave-uniform
#pragma unroll
            for (int r = 0; r < K - 1; ++r) {
#pragma unroll
                for (int d = 0; d < NR; ++d) ring[r][d] = ring[r + 1][d];
            }
            settle(ring[K - 1], nx[u]);
            load_win<1, 2>(nx[u], row(t + PF), b0, g.L);
            us2 sE[ND][5], sO[ND][5], P[ND][10], M[ND - 1][6];
            u32x4 ov;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
#pragma unroll
                for (int r = 0; r < K; ++r) {
                    sE[d][r] = __builtin_bit_cast(us2, ring[r][2 * d]);
                    sO[d][r] = __builtin_bit_cast(us2, ring[r][2 * d + 1]);
                }
                med5_sort5(sE[d]);
                med5_sort5(sO[d]);
                med5_merge(sE[d], sO[d], P[d]);
                if (d >= 1) {                                    // columns 4j .. 4j + 3 | 4j + 2 .. 4j + 5 of dword d - 1
                    us2 Q[10];
#pragma unroll
                    for (int i = 0; i < 10; ++i) Q[i] = hi_lo(P[d - 1][i], P[d][i]);
                    med5_middle(P[d - 1], Q, M[d - 1]);
                }
                if (d >= 2) {                                    // output dword e = d - 1 (window dword index; row bytes 4 (e - 1) ..)
                    const int e = d - 1;
                    us2 ce[5], co[5], me[6];
#pragma unroll
                    for (int r = 0; r < 5; ++r) {
                        ce[r] = hi_lo(sE[e][r], sE[e + 1][r]);   // columns 4j + 2 | 4j + 4 for the windows at 4j | 4j + 2
                        co[r] = hi_lo(sO[e - 1][r], sO[e][r]);   // columns 4j - 1 | 4j + 1 for the windows at 4j + 1 | 4j + 3
                    }
#pragma unroll
                    for (int r = 0; r < 6; ++r) me[r] = hi_lo(M[e - 1][r], M[e][r]);   // columns 4j - 2 .. 4j + 1 | 4j .. 4j + 3
                    const us2 oe = med5_final(me, ce);
                    const us2 oo = med5_final(M[e], co);
                    ov[e - 1] = pack_fields(oe, oo);
                }
            }
            st16_nt(dimg + (int64_t)(y0 + yo) * g.d_row + b0, ov);
        }
    }
}

// ---- 7 x 7 median on the strip walker: the same idea one level up ------------------------------------------------------
// (scripts/gen_median7_net.py: every piece verified on all sorted 0-1 inputs.)  A window is three column pairs and one
// more column: columns sorted once (7 windows share a column), even-start pairs merged once (P_d = pairs (4j, 4j+1) |
// (4j+2, 4j+3) of dword d), the two pairs of a dword merged into a 28-list once (Q_d; two dwords per packed operation:
// P_d.lo | P_d+1.lo against P_d.hi | P_d+1.hi), and of a quad with the pair on its left | on its right the middle eight of
// the 42 once per two windows: T_d = mid8(Q_d | Q_d, P_d-1.hi | P_d+1.lo).  The windows at 4j | 4j+2 then add the columns
// 4j-3 | 4j-1 (the sorted O field of dword d-1 as it is) and those at 4j+1 | 4j+3 the columns 4j+4 | 4j+6 (the sorted E
// field of dword d+1).  Per 16 pixels 12 sorts, 6 merges, 2 quads, 4 mid8, 8 finals = 384 + 252 + 220 + 288 + 112 packed
// min / max and ~250 v_perm / v_alignbit: 94 per pixel, where the per-pixel selection of ipx_rank.hip takes ~400.
#define M7_T us2
#define M7_MIN(a, b) __builtin_elementwise_min(a, b)
#define M7_MAX(a, b) __builtin_elementwise_max(a, b)
#define M7_FN __device__ __forceinline__
#include "ipx_median7_net.inc"
#undef M7_T
#undef M7_MIN
#undef M7_MAX
#undef M7_FN

__device__ __forceinline__ us2 perm2(us2 hi_src, us2 lo_src, uint32_t sel)
{
    return __builtin_bit_cast(us2, __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi_src), __builtin_bit_cast(uint32_t, lo_src), sel));
}

template <int PF>
__global__ __launch_bounds__(256) void median7_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, ConvGeom g)
{
    constexpr int K = 7, ND = 6;                                 // dwords d = 0 .. 5 of the window: row bytes b0 - 4 .. b0 + 19
    typedef Win<1, 3> WR;
    static_assert(WR::ND == ND, "one halo dword on either side");
    unsigned img; int y0, y1, lane0;
    if (!wave_item(g, img, y0, y1, lane0)) return;
    const int gl = lane0 + (int)(threadIdx.x & 63u);
    if (gl >= g.nlanes) return;
    const int b0 = lane_b0(g, gl);
    const bool first = b0 == 0, last = b0 + 16 == g.L;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    const int nrows = y1 - y0;
    auto row = [&](int t) {                                      // walk step t = source row y0 - 3 + t, replicated at the ends
        int yy = y0 - 3 + t;
        yy = yy < 0 ? 0 : yy > g.h - 1 ? g.h - 1 : yy;
        return simg + (int64_t)yy * g.s_row;
    };
    auto even = [](uint32_t w) { return __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, w, 0x0c020c00u)); };   // bytes (0, 2)
    auto odd = [](uint32_t w) { return __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, w, 0x0c030c01u)); };    // bytes (1, 3)
    uint32_t ring[K][ND];                                        // the last 7 source rows, raw (84 field registers would not fit)
    auto settle = [&](uint32_t (&o)[ND], WR& W) {
        if (first) W.w[0] = (W.w[1] & 255u) * 0x01010101u;       // bytes -4 .. -1 = byte 0
        if (last) W.w[5] = (W.w[4] >> 24) * 0x01010101u;         // bytes 16 .. 19 = byte 15
#pragma unroll
        for (int d = 0; d < ND; ++d) o[d] = W.w[d];
    };
    {
        WR w0[K - 1];
#pragma unroll
        for (int j = 0; j < K - 1; ++j) load_win<1, 3>(w0[j], row(j), b0, g.L);
#pragma unroll
        for (int j = 0; j < K - 1; ++j) settle(ring[j + 1], w0[j]);
    }
    WR nx[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) load_win<1, 3>(nx[j], row(K - 1 + j), b0, g.L);
    for (int t0 = K - 1; t0 - (K - 1) < nrows; t0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t = t0 + u, yo = t - (K - 1);
            if (yo >= nrows) break;                              // wave-uniform
#pragma unroll
            for (int r = 0; r < K - 1; ++r) {
#pragma unroll
                for (int d = 0; d < ND; ++d) ring[r][d] = ring[r + 1][d];
            }
            settle(ring[K - 1], nx[u]);
            load_win<1, 3>(nx[u], row(t + PF), b0, g.L);
            us2 sE[ND][7], sO[ND][7], P[ND][14];
            auto pairs = [&](int d) {                            // sorted E / O fields and their merge for dword d
#pragma unroll
                for (int r = 0; r < K; ++r) { sE[d][r] = even(ring[r][d]); sO[d][r] = odd(ring[r][d]); }
                med7_sort7(sE[d]);
                med7_sort7(sO[d]);
                med7_merge(sE[d], sO[d], P[d]);
            };
            u32x4 ov;
            auto outputs = [&](int e) {                          // output dwords e, e + 1 (window dword indices; e = 1 or 3)
                us2 A[14], B[14], QQ[28];
#pragma unroll
                for (int i = 0; i < 14; ++i) {
                    A[i] = perm2(P[e + 1][i], P[e][i], 0x05040100u);     // P_e.lo | P_e+1.lo
                    B[i] = perm2(P[e + 1][i], P[e][i], 0x07060302u);     // P_e.hi | P_e+1.hi
                }
                med7_quad(A, B, QQ);                                     // Q_e | Q_e+1
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int d = e + h;
                    us2 Qd[28], third[14], T[8];
#pragma unroll
                    for (int i = 0; i < 28; ++i) Qd[i] = perm2(QQ[i], QQ[i], h ? 0x03020302u : 0x01000100u);
#pragma unroll
                    for (int i = 0; i < 14; ++i) third[i] = hi_lo(P[d - 1][i], P[d + 1][i]);
                    med7_mid8(Qd, third, T);
                    const us2 oe = med7_final(T, sO[d - 1]);             // windows 4j | 4j + 2 add columns 4j - 3 | 4j - 1
                    const us2 oo = med7_final(T, sE[d + 1]);             // windows 4j + 1 | 4j + 3 add columns 4j + 4 | 4j + 6
                    ov[d - 1] = pack_fields(oe, oo);
                }
            };
            pairs(0); pairs(1); pairs(2); pairs(3);
            outputs(1);
            pairs(4); pairs(5);
            outputs(3);
            st16_nt(dimg + (int64_t)(y0 + yo) * g.d_row + b0, ov);
        }
    }
}

// ---- 9 x 9 median on the strip walker -----------------------------------------------------------------------------------
// (scripts/gen_median9_net.py.)  A window is four column pairs and one more column.  With P_d = the merged pairs
// (4j, 4j+1) | (4j+2, 4j+3) of dword d, ONE packed quad merge per dword gives both alignments of neighbouring pairs,
//   QR_d = quad(P_d, P_d.hi | P_d+1.lo) = quad(pairs 2d, 2d+1) | quad(pairs 2d+1, 2d+2),
// and one packed selection per dword the middle ten of both alignments of neighbouring quads,
//   OO_d = mid10(QR_d-1, QR_d) = columns 4j-4 .. 4j+3 | columns 4j-2 .. 4j+5,
// each of which serves the window that adds the column on its left and the one that adds the column on its right: the
// windows at 4j | 4j+2 are OO_d with the sorted E field of dword d+1 (columns 4j+4 | 4j+6) and those at 4j+1 | 4j+3 are
// OO_d.hi | OO_d+1.lo with the sorted O field of dword d-1 (columns 4j-3 | 4j-1).  No byte shuffles beyond 148 v_alignbit per
// 16 pixels; 12 sorts, 6 merges, 6 quads, 5 mid10, 8 finals = 600 + 360 + 924 + 600 + 144 packed min / max: 164 per pixel.
#define M9_T us2
#define M9_MIN(a, b) __builtin_elementwise_min(a, b)
#define M9_MAX(a, b) __builtin_elementwise_max(a, b)
#define M9_FN __device__ __forceinline__
#include "ipx_median9_net.inc"
#undef M9_T
#undef M9_MIN
#undef M9_MAX
#undef M9_FN

template <int PF>
__global__ __launch_bounds__(256) void median9_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, ConvGeom g)
{
    constexpr int K = 9, ND = 6;                                 // dwords d = 0 .. 5 of the window: row bytes b0 - 4 .. b0 + 19
    typedef Win<1, 4> WR;
    static_assert(WR::ND == ND, "one halo dword on either side");
    unsigned img; int y0, y1, lane0;
    if (!wave_item(g, img, y0, y1, lane0)) return;
    const int gl = lane0 + (int)(threadIdx.x & 63u);
    if (gl >= g.nlanes) return;
    const int b0 = lane_b0(g, gl);
    const bool first = b0 == 0, last = b0 + 16 == g.L;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    const int nrows = y1 - y0;
    auto row = [&](int t) {                                      // walk step t = source row y0 - 4 + t, replicated at the ends
        int yy = y0 - 4 + t;
        yy = yy < 0 ? 0 : yy > g.h - 1 ? g.h - 1 : yy;
        return simg + (int64_t)yy * g.s_row;
    };
    auto even = [](uint32_t w) { return __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, w, 0x0c020c00u)); };   // bytes (0, 2)
    auto odd = [](uint32_t w) { return __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, w, 0x0c030c01u)); };    // bytes (1, 3)
    uint32_t ring[K][ND];                                        // the last 9 source rows, raw
    auto settle = [&](uint32_t (&o)[ND], WR& W) {
        if (first) W.w[0] = (W.w[1] & 255u) * 0x01010101u;       // bytes -4 .. -1 = byte 0
        if (last) W.w[5] = (W.w[4] >> 24) * 0x01010101u;         // bytes 16 .. 19 = byte 15
#pragma unroll
        for (int d = 0; d < ND; ++d) o[d] = W.w[d];
    };
    {
        WR w0[K - 1];
#pragma unroll
        for (int j = 0; j < K - 1; ++j) load_win<1, 4>(w0[j], row(j), b0, g.L);
#pragma unroll
        for (int j = 0; j < K - 1; ++j) settle(ring[j + 1], w0[j]);
    }
    WR nx[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) load_win<1, 4>(nx[j], row(K - 1 + j), b0, g.L);
    for (int t0 = K - 1; t0 - (K - 1) < nrows; t0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t = t0 + u, yo = t - (K - 1);
            if (yo >= nrows) break;                              // wave-uniform
#pragma unroll
            for (int r = 0; r < K - 1; ++r) {
#pragma unroll
                for (int d = 0; d < ND; ++d) ring[r][d] = ring[r + 1][d];
            }
            settle(ring[K - 1], nx[u]);
            load_win<1, 4>(nx[u], row(t + PF), b0, g.L);
            us2 sE[ND][9], sO[ND][9], P[ND][18], QR[ND][36], OO[ND][10];
            auto pairs = [&](int d) {
#pragma unroll
                for (int r = 0; r < K; ++r) { sE[d][r] = even(ring[r][d]); sO[d][r] = odd(ring[r][d]); }
                med9_sort9(sE[d]);
                med9_sort9(sO[d]);
                med9_merge(sE[d], sO[d], P[d]);
            };
            auto quad = [&](int d) {                             // (P_5 has no right neighbour: QR_5's high half is never used)
                us2 X[18];
#pragma unroll
                for (int i = 0; i < 18; ++i) X[i] = hi_lo(P[d][i], P[d + 1 < ND ? d + 1 : d][i]);
                med9_quad(P[d], X, QR[d]);
            };
            auto octet = [&](int d) { med9_mid10(QR[d - 1], QR[d], OO[d]); };
            u32x4 ov;
            auto out = [&](int d) {
                us2 T[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) T[i] = hi_lo(OO[d][i], OO[d + 1][i]);
                const us2 oe = med9_final(OO[d], sE[d + 1]);     // windows 4j | 4j + 2 add columns 4j + 4 | 4j + 6
                const us2 oo = med9_final(T, sO[d - 1]);         // windows 4j + 1 | 4j + 3 add columns 4j - 3 | 4j - 1
                ov[d - 1] = pack_fields(oe, oo);
            };
            pairs(0); pairs(1); quad(0);
            pairs(2); quad(1); octet(1);
            pairs(3); quad(2); octet(2); out(1);
            pairs(4); quad(3); octet(3); out(2);
            pairs(5); quad(4); octet(4); out(3);
            quad(5); octet(5); out(4);
            st16_nt(dimg + (int64_t)(y0 + yo) * g.d_row + b0, ov);
        }
    }
}

// ---- 11 x 11 median on the strip walker ---------------------------------------------------------------------------------
// (scripts/gen_median11_net.py.)  Five column pairs and one more column; the packed forms line up as for 9 x 9:
//   QR_d = quad(P_d, P_d.hi | P_d+1.lo),   U_d = mid12(QR_d-1, QR_d, P_d+1) = columns 4j-4 .. 4j+5 | columns 4j-2 .. 4j+7,
// the windows at 4j | 4j+2 add columns 4j-5 | 4j-3 (O_d-2.hi | O_d-1.lo) and those at 4j+1 | 4j+3 columns 4j+6 | 4j+8
// (E_d+1.hi | E_d+2.lo).  Two halo dwords on either side.  5238 packed min / max per 16 pixels; the working set (two 44-lists
// and a 22-list in flight above a ring of 11 x 8 dwords) lives in VGPRs and AGPRs, one wave per SIMD.
#define M11_T us2
#define M11_MIN(a, b) __builtin_elementwise_min(a, b)
#define M11_MAX(a, b) __builtin_elementwise_max(a, b)
#define M11_FN __device__ __forceinline__
#include "ipx_median11_net.inc"
#undef M11_T
#undef M11_MIN
#undef M11_MAX
#undef M11_FN

__global__ __launch_bounds__(256) void median11_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, ConvGeom g)
{
    constexpr int K = 11, ND = 8;                                // dwords D = 0 .. 7 of the window: row bytes b0 - 8 .. b0 + 23
    typedef Win<1, 5> WR;
    static_assert(WR::ND == ND, "two halo dwords on either side");
    unsigned img; int y0, y1, lane0;
    if (!wave_item(g, img, y0, y1, lane0)) return;
    const int gl = lane0 + (int)(threadIdx.x & 63u);
    if (gl >= g.nlanes) return;
    const int b0 = lane_b0(g, gl);
    const bool first = b0 == 0, last = b0 + 16 == g.L;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    const int nrows = y1 - y0;
    auto row = [&](int t) {                                      // walk step t = source row y0 - 5 + t, replicated at the ends
        int yy = y0 - 5 + t;
        yy = yy < 0 ? 0 : yy > g.h - 1 ? g.h - 1 : yy;
        return simg + (int64_t)yy * g.s_row;
    };
    auto even = [](uint32_t w) { return __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, w, 0x0c020c00u)); };   // bytes (0, 2)
    auto odd = [](uint32_t w) { return __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, w, 0x0c030c01u)); };    // bytes (1, 3)
    uint32_t ring[K][ND];                                        // the last 11 source rows, raw
    auto settle = [&](uint32_t (&o)[ND], WR& W) {
        if (first) W.w[0] = W.w[1] = (W.w[2] & 255u) * 0x01010101u;      // bytes -8 .. -1 = byte 0
        if (last) W.w[6] = W.w[7] = (W.w[5] >> 24) * 0x01010101u;        // bytes 16 .. 23 = byte 15
#pragma unroll
        for (int d = 0; d < ND; ++d) o[d] = W.w[d];
    };
    {
#pragma unroll
        for (int j = 0; j < K - 1; ++j) {
            WR w0;
            load_win<1, 5>(w0, row(j), b0, g.L);
            settle(ring[j + 1], w0);
        }
    }
    WR nx;
    load_win<1, 5>(nx, row(K - 1), b0, g.L);
    for (int t = K - 1; t - (K - 1) < nrows; ++t) {
        const int yo = t - (K - 1);
#pragma unroll
        for (int r = 0; r < K - 1; ++r) {
#pragma unroll
            for (int d = 0; d < ND; ++d) ring[r][d] = ring[r + 1][d];
        }
        settle(ring[K - 1], nx);
        load_win<1, 5>(nx, row(t + 1), b0, g.L);
        us2 sE[ND][11], sO[ND][11], P[ND][22], QR[ND][44], U[ND][12];
        auto sortE = [&](int d) {
            us2 v[11];
#pragma unroll
            for (int r = 0; r < K; ++r) v[r] = even(ring[r][d]);
            med11_sort11(v, sE[d]);
        };
        auto sortO = [&](int d) {
            us2 v[11];
#pragma unroll
            for (int r = 0; r < K; ++r) v[r] = odd(ring[r][d]);
            med11_sort11(v, sO[d]);
        };
        auto pairs = [&](int d) { sortE(d); sortO(d); med11_merge(sE[d], sO[d], P[d]); };
        auto quad = [&](int d) {
            us2 X[22];
#pragma unroll
            for (int i = 0; i < 22; ++i) X[i] = hi_lo(P[d][i], P[d + 1][i]);
            med11_quad(P[d], X, QR[d]);
        };
        u32x4 ov;
        auto out = [&](int d) {                                  // output dword d = 2 .. 5 (row bytes 4 (d - 2) ..)
            med11_mid12(QR[d - 1], QR[d], P[d + 1], U[d]);
            us2 ce[11], co[11];
#pragma unroll
            for (int r = 0; r < 11; ++r) {
                ce[r] = hi_lo(sO[d - 2][r], sO[d - 1][r]);       // columns 4j - 5 | 4j - 3 for the windows at 4j | 4j + 2
                co[r] = hi_lo(sE[d + 1][r], sE[d + 2][r]);       // columns 4j + 6 | 4j + 8 for the windows at 4j + 1 | 4j + 3
            }
            ov[d - 2] = pack_fields(med11_final(U[d], ce), med11_final(U[d], co));
        };
        sortO(0);
        pairs(1); pairs(2); quad(1);
        pairs(3); quad(2);
        pairs(4);
        out(2);
        quad(3); pairs(5); out(3);
        quad(4); pairs(6); out(4);
        quad(5); sortE(7); out(5);
        st16_nt(dimg + (int64_t)(y0 + yo) * g.d_row + b0, ov);
    }
}

// ---- generic two-pass path (any odd K <= IPX_MAX_GAUSS_K) -----------------------------------------------
struct TapsBig { uint16_t k[IPX_MAX_GAUSS_K + 1]; };

// Row pass: tmp[(y + r) * L + b] = sum_j kx[j] * srcrow(y)[reflect(px + j - r) * C + ch], y in [-r, h + r)
__global__ __launch_bounds__(256) void sep_rows_kernel(const uint8_t* __restrict__ src, uint16_t* __restrict__ tmp,
                                                       ConvGeom g, int C, int K, TapsBig kx)
{
    const int r = K / 2;
    const int trow = blockIdx.y;                              // 0 .. h + 2r - 1
    const unsigned img = blockIdx.z;
    const uint8_t* rp = row_ptr(g, src + (int64_t)img * g.s_img, trow - r, r);
    uint16_t* tp = tmp + ((int64_t)img * (g.h + 2 * r) + trow) * (int64_t)g.L;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < g.L; b += gridDim.x * 256) {
        const int px = b / C, ch = b - px * C;
        uint32_t acc = 0;
        if (px >= r && px + r < g.w) {
            const uint8_t* p = rp + b - r * C;
            for (int j = 0; j < K; ++j) acc += (uint32_t)kx.k[j] * p[j * C];
        } else {
            for (int j = 0; j < K; ++j) acc += (uint32_t)kx.k[j] * rp[reflect101(px + j - r, g.w) * C + ch];
        }
        tp[b] = (uint16_t)acc;
    }
}

// Column pass: dst[y][b] = (sum_i ky[i] * tmp[y + i][b] + 2^15) >> 16
__global__ __launch_bounds__(256) void sep_cols_kernel(const uint16_t* __restrict__ tmp, uint8_t* __restrict__ dst,
                                                       ConvGeom g, int K, TapsBig ky)
{
    const int r = K / 2;
    const int y = blockIdx.y;
    const unsigned img = blockIdx.z;
    const uint16_t* tp = tmp + ((int64_t)img * (g.h + 2 * r) + y) * (int64_t)g.L;
    uint8_t* dp = dst + (int64_t)img * g.d_img + (int64_t)y * g.d_row;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < g.L; b += gridDim.x * 256) {
        uint32_t acc = 32768u;
        for (int i = 0; i < K; ++i) acc += (uint32_t)ky.k[i] * tp[(int64_t)i * g.L + b];
        acc >>= 16;
        dp[b] = (uint8_t)(acc > 255u ? 255u : acc);
    }
}

// ---- the two passes on the dot-product instructions (k > 31 with taps <= 255: every Gaussian of that size) ---------------
// The kernels above do one multiply per tap, pixel and pass with a reflected address each: 4.9 ms per 24 x 4K gray at 33 taps
// (31 taps on the matrix cores: 0.1 ms).  Same scratch image, same arithmetic (exact integer sums), but:
//   row pass    : a block stages 1 KiB of a row plus its (k - 1) C halo bytes in LDS (REFLECT_101 applied while staging) and a
//                 lane produces FOUR adjacent outputs: the row taps spread over the byte distance (tap j at byte j C, zeros
//                 between) are read four at a time as one dword (scalar loads from the kernarg), the window advances one
//                 aligned LDS dword per step, and v_alignbyte_b32 gives the three shifted copies: 1 LDS read + 3 + 4 vector
//                 instructions (v_dot4_u32_u8) per 16 multiplies.
//   column pass : a block stages 64 byte columns x (64 + k - 1) rows of the Q8.8 scratch image TRANSPOSED in LDS, vertically
//                 adjacent values packed two per dword ([column][row pair], an odd pitch: conflict-free for a wave's 64
//                 columns); an output is then (k + 1) / 2 LDS dword reads and as many v_dot2_u32_u16 against the tap pairs
//                 (two tables: an odd output row starts in the middle of a pair), rounding constant in the accumulator.
constexpr int kDotChunk = 1024, kDotMaxM = 256, kDotRows = 4;   // a row-pass block: 1 KiB of kDotRows consecutive scratch rows
struct TapsDot4 { int m; uint32_t w[kDotMaxM]; };          // m dwords of four spread row taps each (zero-padded)
// column taps as pairs, three zero pairs on either side: a lane works on FOUR output rows of one parity at once (their windows are
// one pair apart), so that a window dword is read once and meets the tap pairs m, m - 1, m - 2, m - 3
constexpr int kDot2Pad = 3;
struct TapsDot2 { uint32_t even[IPX_MAX_GAUSS_K / 2 + 2 + 2 * kDot2Pad], odd[IPX_MAX_GAUSS_K / 2 + 2 + 2 * kDot2Pad]; };

__global__ __launch_bounds__(256) void sep_rows_dot4_kernel(const uint8_t* __restrict__ src, uint16_t* __restrict__ tmp,
                                                            ConvGeom g, int C, int r, TapsDot4 tp)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dot[];
    const int trow0 = (int)blockIdx.y * kDotRows;             // scratch rows trow0 .. trow0 + kDotRows - 1 of h + 2r
    const int nrows = g.h + 2 * r;
    const unsigned img = blockIdx.z;
    const int chunk0 = (int)blockIdx.x * kDotChunk;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    const int tid = (int)threadIdx.x;
    const int tile_bytes = kDotChunk + 4 * tp.m + 4, o = chunk0 - r * C;          // tile byte t is row byte o + t
    const int tile_dw = tile_bytes >> 2;
    for (int i = tid; i < kDotRows * tile_dw; i += 256) {                          // (kDotRows tiles, one after the other in LDS)
        const int rr = i / tile_dw, t = 4 * (i - rr * tile_dw);
        const int trow = trow0 + rr < nrows ? trow0 + rr : nrows - 1;
        const uint8_t* rp = row_ptr(g, simg, trow - r, r);
        const int rb = o + t;
        uint32_t v = 0;
        if (rb >= 0 && rb + 4 <= g.L) {
            v = *(const u32_u*)(rp + rb);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int b = rb + j;
                const int px = b >= 0 ? b / C : -((-b + C - 1) / C);
                const int ch = b - px * C;
                v |= (uint32_t)rp[reflect101(px, g.w) * C + ch] << (8 * j);
            }
        }
        *(uint32_t*)(s_dot + rr * tile_bytes + t) = v;
    }
    __syncthreads();
    const int off = 4 * tid, b0 = chunk0 + off;
    if (b0 >= g.L) return;
    uint32_t acc[kDotRows][4], d0[kDotRows];
#pragma unroll
    for (int rr = 0; rr < kDotRows; ++rr) {
        acc[rr][0] = acc[rr][1] = acc[rr][2] = acc[rr][3] = 0;
        d0[rr] = *(const uint32_t*)(s_dot + rr * tile_bytes + off);
    }
    for (int mi = 0; mi < tp.m; ++mi) {
        const uint32_t w = tp.w[mi];
#pragma unroll
        for (int rr = 0; rr < kDotRows; ++rr) {
            const uint32_t d1 = *(const uint32_t*)(s_dot + rr * tile_bytes + off + 4 * mi + 4);
            acc[rr][0] = __builtin_amdgcn_udot4(d0[rr], w, acc[rr][0], false);
            acc[rr][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0[rr], 1u), w, acc[rr][1], false);
            acc[rr][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0[rr], 2u), w, acc[rr][2], false);
            acc[rr][3] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0[rr], 3u), w, acc[rr][3], false);
            d0[rr] = d1;
        }
    }
#pragma unroll
    for (int rr = 0; rr < kDotRows; ++rr) {
        if (trow0 + rr >= nrows) break;
        uint16_t* tp_row = tmp + ((int64_t)img * nrows + trow0 + rr) * (int64_t)g.L + b0;
        if (b0 + 4 <= g.L) {
            const u32x2 two = {acc[rr][0] | (acc[rr][1] << 16), acc[rr][2] | (acc[rr][3] << 16)};
            __builtin_memcpy(tp_row, &two, 8);
        } else {
            for (int i = 0; b0 + i < g.L; ++i) tp_row[i] = (uint16_t)acc[rr][i];
        }
    }
}

constexpr int kColW = 64, kColH = 64;
__global__ __launch_bounds__(256) void sep_cols_dot2_kernel(const uint16_t* __restrict__ tmp, uint8_t* __restrict__ dst,
                                                            ConvGeom g, int r, int pitch, TapsDot2 tp)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dot[];
    uint32_t* s_col = reinterpret_cast<uint32_t*>(s_dot);     // [column][row pair], `pitch` dwords per column (odd)
    const int b0 = (int)blockIdx.x * kColW, y0 = (int)blockIdx.y * kColH;
    const unsigned img = blockIdx.z;
    const int rows_out = g.h - y0 < kColH ? g.h - y0 : kColH, rows_in = rows_out + 2 * r;
    const uint16_t* tb = tmp + ((int64_t)img * (g.h + 2 * r) + y0) * (int64_t)g.L;
    const int tid = (int)threadIdx.x;
    const int npairs = (rows_in + 1) >> 1;
    const int col = tid & (kColW - 1);
    const bool col_in = b0 + col < g.L;
#pragma unroll 4
    for (int q = tid >> 6; q < pitch; q += 4) {               // every pair slot of the column is written (zeros past the tile's rows)
        uint32_t lo = 0, hi = 0;
        if (col_in && q < npairs) {
            lo = tb[(int64_t)(2 * q) * g.L + b0 + col];
            if (2 * q + 1 < rows_in) hi = tb[(int64_t)(2 * q + 1) * g.L + b0 + col];
        }
        s_col[col * pitch + q] = lo | (hi << 16);
    }
    __syncthreads();
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (b0 + lane >= g.L) return;
    // wave: output rows of parity (wave & 1) in the tile's upper / lower half; four rows yo, yo + 2, yo + 4, yo + 6 at a time
    const int par = wave & 1;
    const uint32_t* taps = (par ? tp.odd : tp.even) + kDot2Pad;                 // taps[-3 .. -1] and taps[r + 1 .. r + 3] are zero
    const uint32_t* mine = s_col + lane * pitch;
    uint8_t* dp = dst + (int64_t)img * g.d_img + b0 + lane;
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    for (int q0 = 16 * (wave >> 1); q0 < 16 * (wave >> 1) + 16; q0 += 4) {
        if (2 * q0 + par >= rows_out) break;                  // (wave-uniform)
        uint32_t a0 = 32768u, a1 = 32768u, a2 = 32768u, a3 = 32768u;
        const uint32_t* win = mine + q0;
        for (int u = 0; u <= r + 3; ++u) {
            const us2 d = __builtin_bit_cast(us2, win[u]);
            a0 = __builtin_amdgcn_udot2(d, __builtin_bit_cast(us2, taps[u]), a0, false);
            a1 = __builtin_amdgcn_udot2(d, __builtin_bit_cast(us2, taps[u - 1]), a1, false);
            a2 = __builtin_amdgcn_udot2(d, __builtin_bit_cast(us2, taps[u - 2]), a2, false);
            a3 = __builtin_amdgcn_udot2(d, __builtin_bit_cast(us2, taps[u - 3]), a3, false);
        }
        const uint32_t a[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yo = 2 * (q0 + j) + par;
            if (yo < rows_out) {
                const uint32_t v = a[j] >> 16;
                dp[(int64_t)(y0 + yo) * g.d_row] = (uint8_t)(v > 255u ? 255u : v);
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------
// Tuning knobs exist in development builds only (make dev -> lib_dev/libipx.so, -DIPX_DEV); the shipped library reads none.
#ifdef IPX_DEV
int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
int dev_seg_rows() { return env_int("IPX_SEG_ROWS", 0); }   // (development build: re-read on every call, for sweeps)
int dev_wpb() { return env_int("IPX_WPB", 0); }
#else
constexpr int dev_seg_rows() { return 0; }
constexpr int dev_wpb() { return 0; }
#endif

// hd4 = 4 * ceil(r*c/4): halo bytes a main-region lane reads on either side (<= 16)
int make_conv_geom(const char* fn, ConvGeom& g, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c, int r, int max_wpb, int gran,
                   int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                   const uint8_t* top, const uint8_t* bot)
{
    if (int rc = ipx_check_image_args(fn, src, dst, n, h, w, c, c, s_row, s_img, d_row, d_img)) return rc;
    if (r > 0)
        if (int rc = ipx_check_disjoint(fn, "source", src, (int64_t)w * c, s_row, s_img, "destination", dst, (int64_t)w * c, d_row, d_img, n, h)) return rc;
    if ((top || bot) && n != 1) return ipx_fail(IPX_ERR_INVALID, "%s: halo rows need n == 1 (got %d)", fn, n);
    if ((int64_t)w * c > (1 << 30)) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: row of %lld bytes too long", fn, (long long)w * c);
    g.h = h; g.w = w; g.L = w * c;
    g.s_row = s_row; g.s_img = s_img; g.d_row = d_row; g.d_img = d_img;
    g.top = top; g.bot = bot;
#ifdef IPX_DEV
    g.trace = nullptr;
#endif
    const int hd4 = 4 * ((r * c + 3) / 4);
    // lane i covers bytes [16 i, 16 i + 16) and reads hd4 bytes on either side of them; the last two lanes are shifted
    // left to end inside the row (lane_b0), which needs rows of at least 32 + hd4 bytes
    g.nlanes = g.L >= 32 + hd4 ? (g.L + 15) / 16 : 0;
    // Strips are 64 lanes = 1 KiB of the row starting at byte 0, so that every wave's loads and stores cover whole
    // 128-byte lines (measured: strips shifted by 16 bytes stream 10 % slower, scripts/dev/walkbench2.hip).
    g.lanes_per_strip = 64;
    g.nstrips = (g.nlanes + 63) / 64;
    // rows per segment: short enough for >= ~4 waves of work per wave slot, long enough to amortise the
    // 2r halo rows every segment re-reads
    int seg = dev_seg_rows();
    if (seg <= 0) {
        // measured optima (scripts/opbench.py on 16 x 4K, scripts/dev/cfg3ops.py on 128 x 1080p): they do not grow with
        // the workload -- a walker copy streams 5.65 TB/s with 12-row segments, 5.3 with 24, 5.2 with 48
        // (scripts/dev/walkbench3.hip); longer segments only pay where the 2r halo rows are many
        const int hi = 128;
        seg = r <= 1 ? 16 : r == 2 ? 24 : 48;
        seg = (seg + gran / 2) / gran * gran;                 // whole loop iterations (rows per unrolled walk step)
        // between one and two rounds of resident waves (256 CUs x 4 SIMDs x ~4 waves) the second round is a long
        // tail: stretch the segments so that everything is resident at once (gray 4K batches: 55 -> 51 us)
        const int64_t slots = 4096, strips = (int64_t)n * (g.nstrips ? g.nstrips : 1);
        const int64_t waves = strips * ((h + seg - 1) / seg);
        if (waves > slots && waves < 2 * slots) {
            int s2 = (int)((strips * h + slots - 1) / slots);
            s2 = (s2 + gran - 1) / gran * gran;
            if (s2 > seg && s2 <= hi) seg = s2;
        }
        // a launch that cannot fill half the slots (ONE frame per request, a few small ones): a lone wave walks its strip
        // row by row with a handful of loads in flight, so the launch lasts as long as its longest walk, whatever the
        // bandwidth.  Shorter segments, down to one unrolled step, until there are ~2048 waves; the extra halo rows
        // (6 rows read per 4 written at r = 1) hit in L2.  One 4K gray frame, Sobel: 9.3 -> 7.4 us (scripts/dev/onetune.py)
        if (waves < slots / 2) {
            int s2 = (int)(strips * h / (slots / 2)) / gran * gran;
            if (s2 < gran) s2 = gran;
            if (s2 < seg) seg = s2;
        }
    }
    if (seg > h) seg = h;
    g.seg_rows = seg;
    g.nsegs = (h + seg - 1) / seg;
    // waves per block: the strips of one row segment split evenly over as few blocks as max_wpb allows
    int want = dev_wpb();
    if (want <= 0) want = 4;
    if (want > max_wpb) want = max_wpb;
    g.blocks_per_seg = g.nstrips ? (g.nstrips + want - 1) / want : 1;
    g.wpb = g.nstrips ? (g.nstrips + g.blocks_per_seg - 1) / g.blocks_per_seg : 1;
    const uint64_t blocks = (uint64_t)n * g.nsegs * (g.nstrips ? g.blocks_per_seg : 0);
    g.nb = g.nlanes > 0 ? 0 : g.L;
    const uint64_t btotal = (uint64_t)n * h * g.nb;
    if (blocks > 0x3fffffffull || btotal > 0xffffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: too many work items", fn);
    g.main_blocks = (unsigned)blocks;
    g.border_total = (unsigned)btotal;
    return IPX_OK;
}

unsigned conv_block(const ConvGeom& g) { return 64u * (unsigned)g.wpb; }
unsigned conv_grid(const ConvGeom& g) { return g.main_blocks + (g.border_total + conv_block(g) - 1) / conv_block(g); }

template <int OP>
int launch_conv3_band(hipStream_t st, const uint8_t* src, uint8_t* dst, const ConvGeom& g, int n, int c, int band_rows);

template <int OP>
int launch_conv3(const char* fn, void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                 int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img, const uint8_t* top, const uint8_t* bot)
{
    ConvGeom g;
    if (int rc = make_conv_geom(fn, g, src, dst, n, h, w, c, 1, kMaxWavesPerBlock, kConv3U, s_row, s_img, d_row, d_img, top, bot)) return rc;
    const dim3 grid(conv_grid(g)), block(conv_block(g));
    hipStream_t st = (hipStream_t)stream;
    // rows of whole 16-byte lanes, at least one full 4 KiB tile wide (BGR frames): LDS band tiles, as for the Gaussian
    int band = g.L % 16 == 0 && g.nlanes >= 256 ? 8 : 0;
#ifdef IPX_DEV
    band = env_int("IPX_BAND3_ROWS", band);                           // development: A/B (0 = strip walker)
    if (g.L % 16) band = 0;
#endif
    if (band > 0) {
        if (int rc = launch_conv3_band<OP>(st, src, dst, g, n, c, band)) return rc;
        IPX_LAUNCH_CHECK("conv3_band_kernel");
        return IPX_OK;
    }
    switch (c) {
        case 1: hipLaunchKernelGGL((conv3_kernel<OP, 1>), grid, block, 0, st, src, dst, g); break;
        case 2: hipLaunchKernelGGL((conv3_kernel<OP, 2>), grid, block, 0, st, src, dst, g); break;
        case 3: hipLaunchKernelGGL((conv3_kernel<OP, 3>), grid, block, 0, st, src, dst, g); break;
        default: hipLaunchKernelGGL((conv3_kernel<OP, 4>), grid, block, 0, st, src, dst, g); break;
    }
    IPX_LAUNCH_CHECK("conv3_kernel");
    return IPX_OK;
}

template <int C, int K>
void launch_sep_small(hipStream_t st, const uint8_t* src, uint8_t* dst, const ConvGeom& g,
                      const uint16_t* kx, const uint16_t* ky)
{
    Taps<K> t;
    for (int i = 0; i < K; ++i) t.kx[i] = kx[i];
    for (int p = 0; p < (K + 1) / 2; ++p) {
        const uint32_t a = ky[2 * p], b = 2 * p + 1 < K ? ky[2 * p + 1] : 0u, z = 2 * p >= 1 ? ky[2 * p - 1] : 0u;
        t.ce[p] = a | (b << 16);                              // rows (2p, 2p+1) of an even output row's window
        t.co[p] = z | (a << 16);                              // ... of an odd output row's window (shifted by one)
    }
    hipLaunchKernelGGL((sep_small_kernel<C, K>), dim3(conv_grid(g)), dim3(conv_block(g)), 0, st, src, dst, g, t);
}

template <int K>
void launch_sep_small_c(hipStream_t st, const uint8_t* src, uint8_t* dst, const ConvGeom& g, int c,
                        const uint16_t* kx, const uint16_t* ky)
{
    switch (c) {
        case 1: launch_sep_small<1, K>(st, src, dst, g, kx, ky); break;
        case 2: launch_sep_small<2, K>(st, src, dst, g, kx, ky); break;
        case 3: launch_sep_small<3, K>(st, src, dst, g, kx, ky); break;
        default: launch_sep_small<4, K>(st, src, dst, g, kx, ky); break;
    }
}


// Taps of the form g * [1 k1 (k2 k1) 1] with g a power of two (see sep_pow2_kernel): fills `out`, returns false otherwise.
bool pow2_symmetric_taps(const uint16_t* kx, const uint16_t* ky, int k, Pow2Taps* out)
{
    if (k != 3 && k != 5) return false;
    auto split = [&](const uint16_t* t, unsigned* g, unsigned* k1, unsigned* k2) {
        const unsigned g0 = t[0];
        if (g0 == 0 || (g0 & (g0 - 1)) != 0) return false;       // outer tap: a power of two, and every tap a multiple of it
        for (int i = 0; i < k; ++i) if (t[i] != t[k - 1 - i] || t[i] % g0) return false;
        *g = g0; *k1 = t[1] / g0; *k2 = k == 5 ? t[2] / g0 : 0;
        return true;
    };
    unsigned gx, gy, kx1, kx2, ky1, ky2;
    if (!split(kx, &gx, &kx1, &kx2) || !split(ky, &gy, &ky1, &ky2)) return false;
    unsigned m = 0;
    while ((1u << m) < gx * gy) ++m;                             // gx * gy = 2^m (both powers of two)
    if (m > 15) return false;
    const unsigned sx = 256 / gx, sy = 256 / gy;                 // sums of the reduced taps (the taps sum to 256)
    const unsigned rnd = 1u << (15 - m);
    if (255u * sx * sy + rnd > 65535u) return false;
    out->kx1 = kx1; out->kx2 = kx2; out->ky1 = ky1; out->ky2 = ky2; out->rnd = rnd; out->sh = 16 - m;
    return true;
}

template <int K, bool HI8 = false>
void launch_sep_pow2(hipStream_t st, const uint8_t* src, uint8_t* dst, const ConvGeom& g, int c, const Pow2Taps& t)
{
    const dim3 grid(conv_grid(g)), block(conv_block(g));
    switch (c) {
        case 1: hipLaunchKernelGGL((sep_pow2_kernel<1, K, HI8>), grid, block, 0, st, src, dst, g, t); break;
        case 2: hipLaunchKernelGGL((sep_pow2_kernel<2, K, HI8>), grid, block, 0, st, src, dst, g, t); break;
        case 3: hipLaunchKernelGGL((sep_pow2_kernel<3, K, HI8>), grid, block, 0, st, src, dst, g, t); break;
        default: hipLaunchKernelGGL((sep_pow2_kernel<4, K, HI8>), grid, block, 0, st, src, dst, g, t); break;
    }
}

// halo_r: the filter's radius (the tile holds band_rows + 2 halo_r rows).  taller: the caller's kernel takes tiles of up to
// kMaxBandRows output rows instead of `band_rows` (8).
//  * Tile widths are near-equal multiples of 16 lanes (4K BGR: 3 x 240, not 256 + 256 + 208; the block is the width rounded up to
//    whole waves): every tile the same work, and 46.8 instead of 49.9 KB of LDS.
//  * Nine-row tiles where they cost no resident tile per CU (LDS is what limits them): a ninth of the halo rows' row passes and
//    LDS-DMA less.  24 x 4K BGR (scripts/dev/band9.py): Gaussian + gray 268.9 -> 262.6 us, 3 taps + gray 248.4 -> 242.2; 6 x 8K:
//    273.7 -> 263.2; the kernels without the gray phase +-0.3 %.
//  * ONE frame (a single request): a 4K BGR frame is 270 x 3 = 810 eight-row tiles on 256 CUs x 3 resident tiles = 768 slots, i.e.
//    a second round of 42 tiles that lasts as long as the first; 240 x 3 = 720 nine-row tiles fit one round: 12.4 against 13.9 us
//    for the 5-tap Gaussian of one frame (scripts/dev/bandrows.py, B = 1).  Launches of 769 ... 960 eight-row tiles take the
//    shortest taller tile that fits one round, if three of them still fit a CU's LDS.
static int band_tiles_per_cu(int64_t lds_bytes)
{
    const int64_t alloc = (lds_bytes + 1279) / 1280 * 1280;       // (allocation granule observed on this part)
    const int64_t t = 163840 / alloc;
    return (int)(t > 8 ? 8 : t);
}

int make_band_geom(BandGeom& bg, const ConvGeom& g, int n, int band_rows, int halo_r = 2, bool taller = false)
{
    bg.g = g;
    int maxl = kBandLanes;
#ifdef IPX_DEV
    maxl = env_int("IPX_BAND_LANES", kBandLanes);                 // development: narrower tiles (64, 128, 192)
#endif
    bg.ntx = (g.nlanes + maxl - 1) / maxl;
    int round = 16;
#ifdef IPX_DEV
    round = env_int("IPX_BAND_ROUND", round);                     // development: A/B of the width rule (64 = whole waves)
#endif
    bg.tile_lanes = ((g.nlanes + bg.ntx - 1) / bg.ntx + round - 1) / round * round;
    bg.pitch = 16 * bg.tile_lanes + 64;
    if (taller && band_rows == 8) {
        const int occ8 = band_tiles_per_cu((int64_t)(8 + 2 * halo_r) * bg.pitch);
        const int64_t slots = (int64_t)occ8 * 256, per_row_band = (int64_t)n * bg.ntx;
        const int64_t t8 = per_row_band * ((g.h + 7) / 8);
        if (band_tiles_per_cu((int64_t)(9 + 2 * halo_r) * bg.pitch) == occ8) band_rows = 9;
        if (t8 > slots && t8 <= slots + slots / 4) {              // a short second round of 8-row tiles: fit one round instead
            for (int r = 9; r <= kMaxBandRows; ++r) {
                if (per_row_band * ((g.h + r - 1) / r) <= slots && band_tiles_per_cu((int64_t)(r + 2 * halo_r) * bg.pitch) == occ8) {
                    band_rows = r;
                    break;
                }
            }
        }
#ifdef IPX_DEV
        band_rows = env_int("IPX_BAND_ROWS_ALL", band_rows);      // development: the tile height of every kernel that takes taller tiles
#endif
    }
    bg.band_rows = band_rows;
    bg.nbands = (g.h + band_rows - 1) / band_rows;
    const uint64_t nb = (uint64_t)n * bg.nbands * bg.ntx;
    if (nb > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "too many band tiles");
    bg.nblocks = (unsigned)nb;
    return IPX_OK;
}

template <auto kern, class... Args>
void launch_band(hipStream_t st, const BandGeom& bg, int r, const uint8_t* src, uint8_t* dst, Args... args)
{
    const size_t lds = (size_t)(bg.band_rows + 2 * r) * bg.pitch;
    if (lds > 64 * 1024) {                                        // tiles above 64 KiB: raise the kernel's limit, once per kernel
        static std::atomic<uint64_t> raised{0};                   // (template instance) and DEVICE (one process may drive several GPUs)
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t bit = 1ull << (dev & 63);
        if (!(raised.fetch_or(bit) & bit))
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(kern, dim3(bg.nblocks), dim3((bg.tile_lanes + 63) / 64 * 64), lds, st, src, dst, bg, args...);
}

template <int K, bool HI8 = false>
int launch_sep_pow2_band(hipStream_t st, const uint8_t* src, uint8_t* dst, const ConvGeom& g, int n, int c, const Pow2Taps& t, int band_rows)
{
    BandGeom bg;
    if (int rc = make_band_geom(bg, g, n, band_rows, K / 2, true)) return rc;
    switch (c) {
        case 1: launch_band<sep_pow2_band_kernel<1, K, HI8>>(st, bg, K / 2, src, dst, t); break;
        case 2: launch_band<sep_pow2_band_kernel<2, K, HI8>>(st, bg, K / 2, src, dst, t); break;
        case 3: launch_band<sep_pow2_band_kernel<3, K, HI8>>(st, bg, K / 2, src, dst, t); break;
        default: launch_band<sep_pow2_band_kernel<4, K, HI8>>(st, bg, K / 2, src, dst, t); break;
    }
    return IPX_OK;
}

template <int OP>
int launch_conv3_band(hipStream_t st, const uint8_t* src, uint8_t* dst, const ConvGeom& g, int n, int c, int band_rows)
{
    BandGeom bg;
    if (int rc = make_band_geom(bg, g, n, band_rows, 1, true)) return rc;
    switch (c) {
        case 1: launch_band<conv3_band_kernel<OP, 1>>(st, bg, 1, src, dst); break;
        case 2: launch_band<conv3_band_kernel<OP, 2>>(st, bg, 1, src, dst); break;
        case 3: launch_band<conv3_band_kernel<OP, 3>>(st, bg, 1, src, dst); break;
        default: launch_band<conv3_band_kernel<OP, 4>>(st, bg, 1, src, dst); break;
    }
    return IPX_OK;
}

}  // namespace

// Called by ipx_morph_u8 (ipx_rank.hip) for structuring elements the strip walker handles: returns 1 if it launched,
// 0 if the element does not qualify (larger than 5 x 5, more than two distinct row masks, rows too short), < 0 on error.
int ipx_morph_small(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                    int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img, int dilate,
                    const uint32_t* row_bits, int se_rows, int se_cols)
{
    if (se_rows != se_cols || (se_rows != 3 && se_rows != 5)) return 0;
    MorphSE se{};
    int nm = 0;
    for (int i = 0; i < se_rows; ++i) {
        if (!row_bits[i]) { se.sel[i] = 0; continue; }
        int m = -1;
        for (int q = 0; q < nm; ++q) if (se.mask[q] == row_bits[i]) m = q;
        if (m < 0) {
            if (nm == 2) return 0;
            se.mask[nm] = row_bits[i];
            m = nm++;
        }
        se.sel[i] = m + 1;
    }
    const int r = se_rows / 2;
    ConvGeom g;
    if (int rc = make_conv_geom("ipx_morph_u8", g, src, dst, n, h, w, 1, r, 4, 2 * r + 1, s_row, s_img, d_row, d_img, nullptr, nullptr)) return rc;
    if (g.nlanes == 0) return 0;                                 // rows too short for the vector lanes: the tile kernel
    const dim3 grid(conv_grid(g)), block(conv_block(g));
    const uint32_t full = (1u << se_rows) - 1u, centre = 1u << r;
    int shape = nm == 2 && se.mask[0] == centre && se.mask[1] == full ? 1 : nm == 1 && se.mask[0] == full ? 2 : 0;
#ifdef IPX_DEV
    if (env_int("IPX_MORPH_GENERIC", 0)) shape = 0;
#endif
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, block, 0, st, src, dst, g, se); };
    auto pick = [&](auto D, auto RR) {
        constexpr bool d = decltype(D)::value;
        constexpr int rr = decltype(RR)::value;
        if (shape == 1) go(morph_small_kernel<d, rr, 1>);
        else if (shape == 2) go(morph_small_kernel<d, rr, 2>);
        else go(morph_small_kernel<d, rr, 0>);
    };
    typedef std::integral_constant<int, 1> R1;
    typedef std::integral_constant<int, 2> R2;
    if (r == 1) { if (dilate) pick(std::true_type{}, R1{}); else pick(std::false_type{}, R1{}); }
    else { if (dilate) pick(std::true_type{}, R2{}); else pick(std::false_type{}, R2{}); }
    if (hipGetLastError() != hipSuccess) return ipx_fail(IPX_ERR_HIP, "launch of morph_small_kernel failed");
    return 1;
}

// The median walkers are bound by the vector pipe, not by memory, and the larger ones hold few waves per SIMD: on a small
// batch (one 4K frame = 4 strips) the default 24- to 48-row segments leave most SIMDs without a wave.  Choose the segment
// length that minimises rounds of resident waves x rows walked per wave (its rows + the K - 1 rows above them).
static void median_segments(ConvGeom& g, int n, int K, int waves_per_simd, int gran)
{
    if (g.nstrips == 0) return;
    const int64_t slots = 1024 * (int64_t)waves_per_simd, strips = (int64_t)n * g.nstrips;
    if (strips * g.nsegs >= 2 * slots) return;                    // enough waves for the defaults
    int best = g.seg_rows;
    int64_t best_cost = INT64_MAX;
    for (int seg = gran; seg <= g.seg_rows; seg += gran) {
        const int64_t waves = strips * ((g.h + seg - 1) / seg);
        const int64_t cost = ((waves + slots - 1) / slots) * (seg + K - 1);
        if (cost < best_cost) { best_cost = cost; best = seg; }
    }
    g.seg_rows = best;
    g.nsegs = (g.h + best - 1) / best;
    g.main_blocks = (unsigned)((uint64_t)n * g.nsegs * g.blocks_per_seg);
}

// Called by ipx_median_blur_u8 (ipx_rank.hip) for k = 3: returns 1 if the strip walker launched, 0 if rows are too short
// for its vector lanes (the tile kernel takes them), < 0 on error.
int ipx_median3_walker(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                       int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img)
{
    ConvGeom g;
    if (int rc = make_conv_geom("ipx_median_blur_u8", g, src, dst, n, h, w, 1, 1, kMaxWavesPerBlock, 3, s_row, s_img, d_row, d_img, nullptr, nullptr)) return rc;
    if (g.nlanes == 0) return 0;
    hipLaunchKernelGGL(median3_kernel, dim3(conv_grid(g)), dim3(conv_block(g)), 0, st, src, dst, g);
    if (hipGetLastError() != hipSuccess) return ipx_fail(IPX_ERR_HIP, "launch of median3_kernel failed");
    return 1;
}

// Called by ipx_median_blur_u8 for k = 5, same contract.
int ipx_median5_walker(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                       int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img)
{
    constexpr int PF = 2;                                        // rows in flight per lane (3: 2 % slower)
    ConvGeom g;
    if (int rc = make_conv_geom("ipx_median_blur_u8", g, src, dst, n, h, w, 1, 2, 4, PF, s_row, s_img, d_row, d_img, nullptr, nullptr)) return rc;
    if (g.nlanes == 0) return 0;
    median_segments(g, n, 5, 3, PF);
    hipLaunchKernelGGL(median5_kernel<PF>, dim3(conv_grid(g)), dim3(conv_block(g)), 0, st, src, dst, g);
    if (hipGetLastError() != hipSuccess) return ipx_fail(IPX_ERR_HIP, "launch of median5_kernel failed");
    return 1;
}

// Called by ipx_median_blur_u8 for k = 7, same contract.
int ipx_median7_walker(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                       int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img)
{
    constexpr int PF = 2;                                        // rows in flight (1: 618 against 568 us)
    ConvGeom g;
    if (int rc = make_conv_geom("ipx_median_blur_u8", g, src, dst, n, h, w, 1, 3, 4, PF, s_row, s_img, d_row, d_img, nullptr, nullptr)) return rc;
    if (g.nlanes == 0) return 0;
    median_segments(g, n, 7, 2, PF);
    hipLaunchKernelGGL(median7_kernel<PF>, dim3(conv_grid(g)), dim3(conv_block(g)), 0, st, src, dst, g);
    if (hipGetLastError() != hipSuccess) return ipx_fail(IPX_ERR_HIP, "launch of median7_kernel failed");
    return 1;
}

// Called by ipx_median_blur_u8 for k = 9, same contract.
int ipx_median9_walker(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                       int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img)
{
    constexpr int PF = 1;                                        // 2 rows in flight double the 25 KB loop body: 1243 against 1127 us
    ConvGeom g;
    if (int rc = make_conv_geom("ipx_median_blur_u8", g, src, dst, n, h, w, 1, 4, 4, PF, s_row, s_img, d_row, d_img, nullptr, nullptr)) return rc;
    if (g.nlanes == 0) return 0;
    median_segments(g, n, 9, 1, PF);
    hipLaunchKernelGGL(median9_kernel<PF>, dim3(conv_grid(g)), dim3(conv_block(g)), 0, st, src, dst, g);
    if (hipGetLastError() != hipSuccess) return ipx_fail(IPX_ERR_HIP, "launch of median9_kernel failed");
    return 1;
}

// Called by ipx_median_blur_u8 for k = 11, same contract.
int ipx_median11_walker(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                        int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img)
{
    ConvGeom g;
    if (int rc = make_conv_geom("ipx_median_blur_u8", g, src, dst, n, h, w, 1, 5, 4, 1, s_row, s_img, d_row, d_img, nullptr, nullptr)) return rc;
    if (g.nlanes == 0) return 0;
    median_segments(g, n, 11, 1, 1);
    hipLaunchKernelGGL(median11_kernel, dim3(conv_grid(g)), dim3(conv_block(g)), 0, st, src, dst, g);
    if (hipGetLastError() != hipSuccess) return ipx_fail(IPX_ERR_HIP, "launch of median11_kernel failed");
    return 1;
}

extern "C" {

int ipx_laplacian_sharpen_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                             int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                             const uint8_t* top_halo, const uint8_t* bottom_halo)
{
    return launch_conv3<OP_SHARPEN>("ipx_laplacian_sharpen_u8", stream, src, dst, n, h, w, c, s_row, s_img, d_row, d_img,
                                    top_halo, bottom_halo);
}

int ipx_box3_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                const uint8_t* top_halo, const uint8_t* bottom_halo)
{
    return launch_conv3<OP_BOX>("ipx_box3_u8", stream, src, dst, n, h, w, c, s_row, s_img, d_row, d_img, top_halo, bottom_halo);
}

int ipx_sobel_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                 int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                 const uint8_t* top_halo, const uint8_t* bottom_halo)
{
    return launch_conv3<OP_SOBEL>("ipx_sobel_u8", stream, src, dst, n, h, w, c, s_row, s_img, d_row, d_img, top_halo, bottom_halo);
}

int ipx_sobel_lut_u8(void* stream, const uint8_t* src, uint8_t* dst_sobel, uint8_t* dst_lut, int n, int h, int w,
                     int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img, const uint8_t* lut_dev, int64_t lut_stride)
{
    return ipx_conv3_lut_u8(stream, IPX_CONV3_SOBEL, src, dst_sobel, dst_lut, n, h, w, s_row, s_img, d_row, d_img, lut_dev, lut_stride, nullptr, nullptr);
}

int ipx_conv3_lut_u8(void* stream, int op, const uint8_t* src, uint8_t* dst_sobel, uint8_t* dst_lut, int n, int h, int w,
                     int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img, const uint8_t* lut_dev, int64_t lut_stride,
                     const uint8_t* top_halo, const uint8_t* bottom_halo)
{
    const char* fn = "ipx_conv3_lut_u8";
    ipx_clear_error();
    if (op != IPX_CONV3_SHARPEN && op != IPX_CONV3_BOX && op != IPX_CONV3_SOBEL)
        return ipx_fail(IPX_ERR_INVALID, "%s: op=%d (0 = Laplacian sharpening, 1 = 3x3 box, 2 = Sobel)", fn, op);
    if (!lut_dev) return ipx_fail(IPX_ERR_INVALID, "%s: null table", fn);
    if (lut_stride != 0 && lut_stride < 256) return ipx_fail(IPX_ERR_INVALID, "%s: lut_stride %lld must be 0 or >= 256", fn, (long long)lut_stride);
    if (int rc = ipx_check_image_args(fn, src, dst_lut, n, h, w, 1, 1, s_row, s_img, d_row, d_img)) return rc;
    if (int rc = ipx_check_disjoint(fn, "the Sobel destination", dst_sobel, w, d_row, d_img, "the table destination", dst_lut, w, d_row, d_img, n, h)) return rc;
    if (int rc = ipx_check_disjoint(fn, "source", src, w, s_row, s_img, "the table destination", dst_lut, w, d_row, d_img, n, h)) return rc;
    ConvGeom g;
    if (int rc = make_conv_geom(fn, g, src, dst_sobel, n, h, w, 1, 1, kMaxWavesPerBlock, kConv3U, s_row, s_img, d_row, d_img, top_halo, bottom_halo)) return rc;
    const dim3 grid(conv_grid(g)), block(conv_block(g));
    hipStream_t st = (hipStream_t)stream;
    if (op == IPX_CONV3_SOBEL) hipLaunchKernelGGL((conv3_lut_kernel<OP_SOBEL>), grid, block, 0, st, src, dst_sobel, dst_lut, g, lut_dev, lut_stride);
    else if (op == IPX_CONV3_BOX) hipLaunchKernelGGL((conv3_lut_kernel<OP_BOX>), grid, block, 0, st, src, dst_sobel, dst_lut, g, lut_dev, lut_stride);
    else hipLaunchKernelGGL((conv3_lut_kernel<OP_SHARPEN>), grid, block, 0, st, src, dst_sobel, dst_lut, g, lut_dev, lut_stride);
    IPX_LAUNCH_CHECK("conv3_lut_kernel");
    return IPX_OK;
}

int ipx_sep_filter_q8_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                         int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                         const uint16_t* kx_q8, const uint16_t* ky_q8, int k,
                         const uint8_t* top_halo, const uint8_t* bottom_halo)
{
    const char* fn = "ipx_sep_filter_q8_u8";
    ipx_clear_error();
    if (!kx_q8 || !ky_q8) return ipx_fail(IPX_ERR_INVALID, "%s: null taps", fn);
    if (k < 1 || (k & 1) == 0 || k > IPX_MAX_GAUSS_K) return ipx_fail(IPX_ERR_INVALID, "%s: k=%d must be odd and in 1..%d", fn, k, IPX_MAX_GAUSS_K);
    ConvGeom g;
    if (int rc = make_conv_geom(fn, g, src, dst, n, h, w, c, k / 2, k <= 5 ? kMaxWavesPerBlock : 8, k <= 7 ? k + 1 : 1, s_row, s_img, d_row, d_img, top_halo, bottom_halo)) return rc;
    unsigned sx = 0, sy = 0;
    for (int i = 0; i < k; ++i) {
        if (kx_q8[i] > 256 || ky_q8[i] > 256) return ipx_fail(IPX_ERR_INVALID, "%s: tap %d exceeds 256 (Q0.8 of 1.0)", fn, i);
        sx += kx_q8[i]; sy += ky_q8[i];
    }
    if (sx != 256 || sy != 256) return ipx_fail(IPX_ERR_INVALID, "%s: taps must sum to 256 (got %u, %u)", fn, sx, sy);
    hipStream_t st = (hipStream_t)stream;
    if (k == 1) {                                                     // the single tap is 256 = 1.0: a copy (cv2 too)
        uint8_t ident[256];
        for (int i = 0; i < 256; ++i) ident[i] = (uint8_t)i;
        return ipx_lut_u8(stream, src, dst, n, h, w, c, s_row, s_img, d_row, d_img, ident);
    }
    Pow2Taps pt;
    if (g.nlanes > 0 && pow2_symmetric_taps(kx_q8, ky_q8, k, &pt)
#ifdef IPX_DEV
        && !getenv("IPX_NO_POW2")                                     // A/B switch of development builds
#endif
    ) {                                                               // cv2.GaussianBlur's own 3- and 5-tap kernels
        // rows of whole 16-byte lanes, at least one full 4 KiB tile wide: LDS band tiles (5-10 % faster than the strip
        // walker on BGR frames: 229 vs 248 us for 5 taps, 206 vs 232 us for 3 taps per 24 x 4K, scripts/dev/bandab.py)
        int band = g.L % 16 == 0 && g.nlanes >= 256 ? 8 : 0;   // 3 blocks of 41 KiB (3 taps) / 49 KiB (5 taps) per CU; 12-row tiles (66 KiB,
                                                               // 2 blocks) are as fast alone but 1 % slower inside the step
        if (k == 5) {
            // 5 taps gain little (2-5 % on 4K / 8K rows) and only when the equal-width tiles fill the block's four waves:
            // on 1080p BGR rows (2 tiles of 180 lanes) the walker is as fast per launch and 2-4 % faster in a pipeline
            const int ntx = (g.nlanes + kBandLanes - 1) / kBandLanes;
            if ((g.nlanes + ntx - 1) / ntx < 224) band = 0;           // mean tile width in lanes
        }
#ifdef IPX_DEV
        band = env_int("IPX_BAND_ROWS", band);                        // development: A/B (0 = strip walker)
        if (g.L % 16) band = 0;
#endif
        if (band > 0) {
            if (int rc = k == 3 ? launch_sep_pow2_band<3>(st, src, dst, g, n, c, pt, band)
                       : pt.sh == 8 ? launch_sep_pow2_band<5, true>(st, src, dst, g, n, c, pt, band)
                                    : launch_sep_pow2_band<5>(st, src, dst, g, n, c, pt, band)) return rc;
            IPX_LAUNCH_CHECK("sep_pow2_band_kernel");
            return IPX_OK;
        }
        if (k == 3) launch_sep_pow2<3>(st, src, dst, g, c, pt);
        else if (pt.sh == 8) launch_sep_pow2<5, true>(st, src, dst, g, c, pt);     // (cv2's own 5-tap kernel: result = high bytes)
        else launch_sep_pow2<5>(st, src, dst, g, c, pt);
        IPX_LAUNCH_CHECK("sep_pow2_kernel");
        return IPX_OK;
    }
    if (k == 3) { launch_sep_small_c<3>(st, src, dst, g, c, kx_q8, ky_q8); IPX_LAUNCH_CHECK("sep_small_kernel<3>"); return IPX_OK; }
    if (k == 5) {
#ifdef IPX_DEV
        const char* tf = getenv("IPX_CONV_TRACE");                    // dev build: dump per-wave start / end stamps (100 MHz)
        if (tf) {
            const size_t nw = (size_t)conv_grid(g) * g.wpb, bytes = 2 * nw * sizeof(unsigned long long);
            IPX_HIP(hipMalloc((void**)&g.trace, bytes));
            IPX_HIP(hipMemset(g.trace, 0, bytes));
            launch_sep_small_c<5>(st, src, dst, g, c, kx_q8, ky_q8);
            IPX_HIP(hipDeviceSynchronize());
            unsigned long long* hst = (unsigned long long*)malloc(bytes);
            IPX_HIP(hipMemcpy(hst, g.trace, bytes, hipMemcpyDeviceToHost));
            if (FILE* f = fopen(tf, "wb")) { const unsigned long long hdr[2] = {(unsigned long long)g.main_blocks * g.wpb, 0}; fwrite(hdr, 1, sizeof(hdr), f); fwrite(hst, 1, bytes, f); fclose(f); }
            free(hst);
            IPX_HIP(hipFree(g.trace));
            return IPX_OK;
        }
#endif
        launch_sep_small_c<5>(st, src, dst, g, c, kx_q8, ky_q8); IPX_LAUNCH_CHECK("sep_small_kernel<5>"); return IPX_OK;
    }
    if (k == 7) { launch_sep_small_c<7>(st, src, dst, g, c, kx_q8, ky_q8); IPX_LAUNCH_CHECK("sep_small_kernel<7>"); return IPX_OK; }

    if (k >= 9 && k <= 31)                                          // int8 MFMA kernel, ipx_sepmfma.hip (VALU fallback: ipx_sepbig.hip)
        return ipx_sep_mfma(st, src, dst, n, h, w, c, s_row, s_img, d_row, d_img, kx_q8, ky_q8, k, top_halo, bottom_halo);

    // generic two-pass path through a Q8.8 scratch image of (h + 2r) rows per image (k > 31)
    if (n > 65535 || h + k > 65535) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: n or h too large for the two-pass path", fn);
    const int r = k / 2;
    const size_t tmp_bytes = (size_t)n * (size_t)(h + 2 * r) * (size_t)g.L * sizeof(uint16_t);
    uint16_t* tmp = nullptr;
    IPX_HIP(hipMallocAsync((void**)&tmp, tmp_bytes, st));
    TapsBig tx{}, ty{};
    for (int i = 0; i < k; ++i) { tx.k[i] = kx_q8[i]; ty.k[i] = ky_q8[i]; }
    int rc = IPX_OK;
    bool small_taps = true;                                     // every tap fits a byte (any Gaussian of this size; a tap of 256 does not)
    for (int i = 0; i < k; ++i) small_taps = small_taps && kx_q8[i] <= 255 && ky_q8[i] <= 255;
    const int ke = (k - 1) * c + 1;                             // row taps spread over the byte distance
    if (small_taps && (ke + 3) / 4 <= kDotMaxM && g.L <= 0x3fffffff) {
        TapsDot4 t4{};
        t4.m = (ke + 3) / 4;
        for (int j = 0; j < k; ++j) t4.w[(j * c) >> 2] |= (uint32_t)kx_q8[j] << (8 * ((j * c) & 3));
        TapsDot2 t2{};
        for (int m = 0; m <= r; ++m) {
            t2.even[kDot2Pad + m] = (uint32_t)ky_q8[2 * m] | (2 * m + 1 < k ? (uint32_t)ky_q8[2 * m + 1] << 16 : 0u);
            t2.odd[kDot2Pad + m] = (m ? (uint32_t)ky_q8[2 * m - 1] : 0u) | ((uint32_t)ky_q8[2 * m] << 16);
        }
        const size_t lds_rows = (size_t)kDotRows * (kDotChunk + 4 * t4.m + 4) + 16;
        hipLaunchKernelGGL(sep_rows_dot4_kernel, dim3((unsigned)((g.L + kDotChunk - 1) / kDotChunk), (unsigned)((h + 2 * r + kDotRows - 1) / kDotRows), (unsigned)n), dim3(256),
                           lds_rows, st, src, tmp, g, c, r, t4);
        if (hipGetLastError() != hipSuccess) rc = ipx_fail(IPX_ERR_HIP, "%s: launch of sep_rows_dot4_kernel failed", fn);
        if (rc == IPX_OK) {
            const int pitch = ((kColH + 2 * r + 1) / 2 + 4) | 1;       // pairs of the tile's rows + the three a four-row group reads beyond
            hipLaunchKernelGGL(sep_cols_dot2_kernel, dim3((unsigned)((g.L + kColW - 1) / kColW), (unsigned)((h + kColH - 1) / kColH), (unsigned)n), dim3(256),
                               (size_t)kColW * pitch * sizeof(uint32_t), st, tmp, dst, g, r, pitch, t2);
            if (hipGetLastError() != hipSuccess) rc = ipx_fail(IPX_ERR_HIP, "%s: launch of sep_cols_dot2_kernel failed", fn);
        }
        hipError_t e2 = hipFreeAsync(tmp, st);
        if (e2 != hipSuccess && rc == IPX_OK) rc = ipx_fail(IPX_ERR_HIP, "%s: hipFreeAsync failed: %s", fn, hipGetErrorString(e2));
        return rc;
    }
    unsigned gx = (unsigned)((g.L + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(sep_rows_kernel, dim3(gx, (unsigned)(h + 2 * r), (unsigned)n), dim3(256), 0, st, src, tmp, g, c, k, tx);
    if (hipGetLastError() != hipSuccess) rc = ipx_fail(IPX_ERR_HIP, "%s: launch of sep_rows_kernel failed", fn);
    if (rc == IPX_OK) {
        hipLaunchKernelGGL(sep_cols_kernel, dim3(gx, (unsigned)h, (unsigned)n), dim3(256), 0, st, tmp, dst, g, k, ty);
        if (hipGetLastError() != hipSuccess) rc = ipx_fail(IPX_ERR_HIP, "%s: launch of sep_cols_kernel failed", fn);
    }
    hipError_t e = hipFreeAsync(tmp, st);
    if (e != hipSuccess && rc == IPX_OK) rc = ipx_fail(IPX_ERR_HIP, "%s: hipFreeAsync failed: %s", fn, hipGetErrorString(e));
    return rc;
}

// the shapes and kernel sizes ipx_gaussian_blur_gray_u8 serves with ONE launch: rows of whole 16-byte lanes, >= 4 KiB, a row's band
// tiles at least 176 lanes wide for 5 taps.  (The Gaussian ALONE takes band tiles from 224 lanes up -- on 1080p BGR rows, two tiles
// of 180 lanes, its strip walker is 3-5 % faster -- but with the gray image from the same read the band form wins there too:
// 128 x 1080p, Gaussian + gray + histogram + tables 432 us against 516 us for the walker and cvtColor + histogram, scripts/dev/gray1080.py)
static bool gaussian_gray_one_pass(int w, int k)
{
    if (k != 3 && k != 5) return false;
    const int64_t L = 3ll * w;
    if (w <= 0 || L > (1 << 30) || L % 16 != 0) return false;
    const int nlanes = (int)(L / 16);
    const int ntx = (nlanes + kBandLanes - 1) / kBandLanes;
    int min5 = 176;
#ifdef IPX_DEV
    min5 = env_int("IPX_GRAY_ONEPASS_MIN", 176);                     // development: A/B of the 5-tap form's narrowest tiles
#endif
    return nlanes >= 256 && (k == 3 || (nlanes + ntx - 1) / ntx >= min5);
}

int ipx_gaussian_blur_gray_one_pass(int h, int w, int k) { return h > 0 && gaussian_gray_one_pass(w, k) ? 1 : 0; }

int ipx_gaussian_blur_gray_u8(void* stream, const uint8_t* src, uint8_t* dst, uint8_t* gray, int n, int h, int w,
                              int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                              int64_t gray_row_stride, int64_t gray_img_stride, int k)
{
    const char* fn = "ipx_gaussian_blur_gray_u8";
    ipx_clear_error();
    if (int rc = ipx_check_image_args(fn, src, gray, n, h, w, 3, 1, s_row, s_img, gray_row_stride, gray_img_stride)) return rc;
    if (int rc = ipx_check_disjoint(fn, "source", src, 3ll * w, s_row, s_img, "the gray destination", gray, w, gray_row_stride, gray_img_stride, n, h)) return rc;
    if (dst)
        if (int rc = ipx_check_disjoint(fn, "the colour destination", dst, 3ll * w, d_row, d_img, "the gray destination", gray, w, gray_row_stride, gray_img_stride, n, h)) return rc;
    uint16_t q[IPX_MAX_GAUSS_K];
    if (int rc = ipx_gauss_kernel_q8(k, 0.0, q)) return rc;
    ConvGeom g;
    Pow2Taps pt;
    if ((k == 3 || k == 5) && pow2_symmetric_taps(q, q, k, &pt)) {
        if (int rc = make_conv_geom(fn, g, src, dst, n, h, w, 3, k / 2, kMaxWavesPerBlock, k + 1, s_row, s_img, d_row, d_img, nullptr, nullptr)) return rc;
        if (gaussian_gray_one_pass(w, k)) {
            BandGeom bg;
            if (int rc = make_band_geom(bg, g, n, 8, k / 2, true)) return rc;
            const GrayOut go{gray, gray_row_stride, gray_img_stride};
            if (k == 3) launch_band<sep_pow2_band_gray_kernel<3>>((hipStream_t)stream, bg, 1, src, dst, pt, go);
            else if (pt.sh == 8) launch_band<sep_pow2_band_gray_kernel<5, true>>((hipStream_t)stream, bg, 2, src, dst, pt, go);
            else launch_band<sep_pow2_band_gray_kernel<5>>((hipStream_t)stream, bg, 2, src, dst, pt, go);
            IPX_LAUNCH_CHECK("sep_pow2_band_gray_kernel");
            return IPX_OK;
        }
    }
    // every other shape or kernel size: the two entry points one after the other
    if (int rc = ipx_gaussian_blur_u8(stream, src, dst, n, h, w, 3, s_row, s_img, d_row, d_img, k, nullptr, nullptr)) return rc;
    return ipx_bgr2gray_u8(stream, src, gray, n, h, w, 3, s_row, s_img, gray_row_stride, gray_img_stride);
}

int ipx_gaussian_sharpen_gray_u8(void* stream, const uint8_t* src, uint8_t* dst_gauss, uint8_t* dst_sharp, uint8_t* gray, int n, int h, int w,
                                 int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                                 int64_t gray_row_stride, int64_t gray_img_stride, int k)
{
    const char* fn = "ipx_gaussian_sharpen_gray_u8";
    ipx_clear_error();
    if (int rc = ipx_check_image_args(fn, src, gray, n, h, w, 3, 1, s_row, s_img, gray_row_stride, gray_img_stride)) return rc;
    if (int rc = ipx_check_image_args(fn, src, dst_sharp, n, h, w, 3, 3, s_row, s_img, d_row, d_img)) return rc;
    if (int rc = ipx_check_disjoint(fn, "source", src, 3ll * w, s_row, s_img, "the sharpened destination", dst_sharp, 3ll * w, d_row, d_img, n, h)) return rc;
    if (int rc = ipx_check_disjoint(fn, "source", src, 3ll * w, s_row, s_img, "the gray destination", gray, w, gray_row_stride, gray_img_stride, n, h)) return rc;
    if (int rc = ipx_check_disjoint(fn, "the sharpened destination", dst_sharp, 3ll * w, d_row, d_img, "the gray destination", gray, w, gray_row_stride, gray_img_stride, n, h)) return rc;
    if (dst_gauss) {
        if (int rc = ipx_check_disjoint(fn, "the blurred destination", dst_gauss, 3ll * w, d_row, d_img, "the sharpened destination", dst_sharp, 3ll * w, d_row, d_img, n, h)) return rc;
        if (int rc = ipx_check_disjoint(fn, "the blurred destination", dst_gauss, 3ll * w, d_row, d_img, "the gray destination", gray, w, gray_row_stride, gray_img_stride, n, h)) return rc;
    }
    uint16_t q[IPX_MAX_GAUSS_K];
    if (int rc = ipx_gauss_kernel_q8(k, 0.0, q)) return rc;
    ConvGeom g;
    Pow2Taps pt;
    if (k == 5 && pow2_symmetric_taps(q, q, k, &pt) && gaussian_gray_one_pass(w, k)) {
        if (int rc = make_conv_geom(fn, g, src, dst_gauss, n, h, w, 3, 2, kMaxWavesPerBlock, k + 1, s_row, s_img, d_row, d_img, nullptr, nullptr)) return rc;
        BandGeom bg;
        if (int rc = make_band_geom(bg, g, n, 8, 2, true)) return rc;
        if (bg.band_rows != 8 && bg.band_rows != 9)                        // (the kernel's walk is unrolled per tile height: 8 and 9 exist)
            if (int rc = make_band_geom(bg, g, n, 8, 2, false)) return rc;
        const GrayOut go{gray, gray_row_stride, gray_img_stride};
        const SharpOut so{dst_sharp, d_row, d_img};
        hipStream_t st = (hipStream_t)stream;
        if (pt.sh == 8 && bg.band_rows == 9) launch_band<gauss5_sharpen_gray_band_kernel<true, 9>>(st, bg, 2, src, dst_gauss, pt, go, so);
        else if (pt.sh == 8) launch_band<gauss5_sharpen_gray_band_kernel<true, 8>>(st, bg, 2, src, dst_gauss, pt, go, so);
        else if (bg.band_rows == 9) launch_band<gauss5_sharpen_gray_band_kernel<false, 9>>(st, bg, 2, src, dst_gauss, pt, go, so);
        else launch_band<gauss5_sharpen_gray_band_kernel<false, 8>>(st, bg, 2, src, dst_gauss, pt, go, so);
        IPX_LAUNCH_CHECK("gauss5_sharpen_gray_band_kernel");
        return IPX_OK;
    }
    // every other shape or kernel size: the entry points one after the other
    if (int rc = ipx_gaussian_blur_gray_u8(stream, src, dst_gauss, gray, n, h, w, s_row, s_img, d_row, d_img, gray_row_stride, gray_img_stride, k)) return rc;
    return ipx_laplacian_sharpen_u8(stream, src, dst_sharp, n, h, w, 3, s_row, s_img, d_row, d_img, nullptr, nullptr);
}

int ipx_gaussian_blur_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                         int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img, int k,
                         const uint8_t* top_halo, const uint8_t* bottom_halo)
{
    uint16_t q[IPX_MAX_GAUSS_K];
    if (int rc = ipx_gauss_kernel_q8(k, 0.0, q)) return rc;          // rejects k outside 1..IPX_MAX_GAUSS_K before q is written
    return ipx_sep_filter_q8_u8(stream, src, dst, n, h, w, c, s_row, s_img, d_row, d_img, q, q, k, top_halo, bottom_halo);
}

}  // extern "C"

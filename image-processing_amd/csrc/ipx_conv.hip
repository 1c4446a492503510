// ipx_conv.hip -- 3x3 convolutions (Laplacian sharpen, box, Sobel) and the separable fixed-point
// Gaussian of cv2.GaussianBlur, for 1..4 interleaved uint8 channels, BORDER_REFLECT_101.
//
// Work decomposition ("strip walker"): an image row is L = w*c bytes.  One WAVE owns a strip of 64 lanes x
// 16 bytes = 1 KiB of the row and walks down a segment of rows.  Per input row a lane loads its own 16 bytes
// plus the few halo bytes left and right of them (same cache lines its neighbour lanes load: L1 hits), keeps
// the vertical window in registers, and stores 16 output bytes per row.  Every input byte is fetched from HBM
// once (plus r halo rows per segment); nothing is staged twice.  Neighbouring strips are consecutive work items
// and the block->item map is XCD-aware so that they share one L2.
// Kernels larger than 7 taps go through the generic two-pass path at the bottom (row pass to a Q8.8 scratch
// image, column pass back to uint8).
#include "ipx_common.h"
#include <vector>

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = 64 * kWavesPerBlock;
constexpr int kStripBytes = 64 * 16;

struct ConvGeom {
    int h, w, L;                       // rows, pixels per row, bytes per row
    int64_t s_row, s_img, d_row, d_img;
    int nstrips, seg_rows, nsegs;
    unsigned nitems;                   // n * nsegs * nstrips
    const uint8_t* top;                // r halo rows above row 0 (row -r first), or null = reflect
    const uint8_t* bot;                // r halo rows below row h-1, or null = reflect
};

template <int C, int R>
struct Win {                            // bytes [b0 - 4*HD, b0 + 16 + 4*HD) of one row, as dwords
    static constexpr int HD = (R * C + 3) / 4;
    static constexpr int ND = 4 + 2 * HD;
    uint32_t w[ND];
    __device__ __forceinline__ uint32_t at(int i) const   // byte at offset i from b0; i static after unroll
    {
        const int k = i + 4 * HD;
        return (w[k >> 2] >> (8 * (k & 3))) & 255u;
    }
};

// Source row `y` of the image (y may lie outside [0,h)): halo row if provided, else REFLECT_101.
template <int R>
__device__ __forceinline__ const uint8_t* row_ptr(const ConvGeom& g, const uint8_t* simg, int y)
{
    if (y < 0 && g.top) return g.top + (int64_t)(R + y) * g.s_row;
    if (y >= g.h && g.bot) return g.bot + (int64_t)(y - g.h) * g.s_row;
    return simg + (int64_t)reflect101(y, g.h) * g.s_row;
}

// Per-lane byte offsets (from the row pointer) of the window bytes of a lane that touches the left/right image
// border or the ragged end of the row: REFLECT_101 in pixel space.  Row-independent, so computed once per item.
template <int C, int R>
struct Gather {
    int o[Win<C, R>::ND * 4];
    __device__ __forceinline__ void init(int b0, int wpx)
    {
        constexpr int HD = Win<C, R>::HD;
#pragma unroll
        for (int i = 0; i < Win<C, R>::ND * 4; ++i) {
            const int b = b0 + i - 4 * HD;
            int px = (b + 64 * C) / C - 64;                 // floor division, b >= -64*C
            const int ch = b - px * C;
            o[i] = reflect101(px, wpx) * C + ch;
        }
    }
};

template <int C, int R>
__device__ __forceinline__ void load_win_fast(Win<C, R>& W, const uint8_t* rp, int b0)
{
    constexpr int HD = Win<C, R>::HD;
    __builtin_memcpy(&W.w[0], rp + b0 - 4 * HD, 4 * HD);
    __builtin_memcpy(&W.w[HD], rp + b0, 16);
    __builtin_memcpy(&W.w[HD + 4], rp + b0 + 16, 4 * HD);
}

template <int C, int R>
__device__ __forceinline__ void load_win_edge(Win<C, R>& W, const uint8_t* rp, int b0, bool interior, const Gather<C, R>& G)
{
    if (interior) {
        load_win_fast<C, R>(W, rp, b0);
    } else {
#pragma unroll
        for (int d = 0; d < Win<C, R>::ND; ++d) {
            uint32_t word = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) word |= (uint32_t)rp[G.o[4 * d + k]] << (8 * k);
            W.w[d] = word;
        }
    }
}

__device__ __forceinline__ void store_out(uint8_t* dp, int b0, int L, u32x4 o)
{
    if (b0 + 16 <= L) {
        st16(dp + b0, o);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (b0 + i < L) dp[b0 + i] = (uint8_t)(o[i >> 2] >> (8 * (i & 3)));
    }
}

__device__ __forceinline__ uint32_t clamp255(int v) { return (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

enum { OP_SHARPEN = 0, OP_BOX = 1, OP_SOBEL = 2 };

template <int OP, int C>
__device__ __forceinline__ u32x4 conv3_row(const Win<C, 1>& a, const Win<C, 1>& b, const Win<C, 1>& c)
{
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * k + j;
            uint32_t v;
            if (OP == OP_SHARPEN) {          // clip(5c - n - s - e - w)            server.py:270-276
                const int s = (int)(a.at(i) + c.at(i) + b.at(i - C) + b.at(i + C));
                v = clamp255(5 * (int)b.at(i) - s);
            } else if (OP == OP_BOX) {       // round(sum9 / 9): never a tie -> (sum + 4) / 9
                const uint32_t s = a.at(i - C) + a.at(i) + a.at(i + C) + b.at(i - C) + b.at(i) + b.at(i + C) +
                                   c.at(i - C) + c.at(i) + c.at(i + C);
                v = (s + 4u) / 9u;
            } else {                         // min(255, |gx| + |gy|)
                const int l = (int)(a.at(i - C) + 2 * b.at(i - C) + c.at(i - C));
                const int r = (int)(a.at(i + C) + 2 * b.at(i + C) + c.at(i + C));
                const int t = (int)(a.at(i - C) + 2 * a.at(i) + a.at(i + C));
                const int u = (int)(c.at(i - C) + 2 * c.at(i) + c.at(i + C));
                const int gx = r - l, gy = u - t;
                const int m = (gx < 0 ? -gx : gx) + (gy < 0 ? -gy : gy);
                v = (uint32_t)(m > 255 ? 255 : m);
            }
            word |= v << (8 * j);
        }
        o[k] = word;
    }
    return o;
}

// Decode the wave's work item.  Returns false if the wave has nothing to do.
__device__ __forceinline__ bool wave_item(const ConvGeom& g, unsigned& img, int& y0, int& y1, int& sb0)
{
    const unsigned wave = threadIdx.x >> 6;
    const unsigned item = xcd_remap(blockIdx.x, gridDim.x) * kWavesPerBlock + wave;
    if (item >= g.nitems) return false;
    const unsigned strip = item % (unsigned)g.nstrips;
    const unsigned rest = item / (unsigned)g.nstrips;
    const unsigned seg = rest % (unsigned)g.nsegs;
    img = rest / (unsigned)g.nsegs;
    y0 = (int)seg * g.seg_rows;
    y1 = y0 + g.seg_rows < g.h ? y0 + g.seg_rows : g.h;
    sb0 = (int)(strip * kStripBytes);                       // first byte of the wave's strip
    return true;
}

// EDGE = false: every lane of the wave is interior (vector loads only, U rows in flight).
// EDGE = true : the strip touches the left/right border or the ragged row end; lanes that do gather bytes.
template <int OP, int C, bool EDGE>
__device__ __forceinline__ void conv3_walk(const ConvGeom& g, const uint8_t* simg, uint8_t* dimg, int y0, int y1, int b0)
{
    typedef Win<C, 1> W1;
    constexpr int U = EDGE ? 1 : 4;                        // rows loaded ahead per step
    const bool interior = !EDGE || (b0 >= 4 * W1::HD && b0 + 16 + 4 * W1::HD <= g.L);
    Gather<C, 1> G;
    if (EDGE && !interior) G.init(b0, g.w);
    auto load = [&](W1& W, int y) {
        const uint8_t* rp = row_ptr<1>(g, simg, y);
        if (EDGE) load_win_edge<C, 1>(W, rp, b0, interior, G);
        else load_win_fast<C, 1>(W, rp, b0);
    };
    W1 a, b;
    load(a, y0 - 1);
    load(b, y0);
    for (int y = y0; y < y1; y += U) {
        W1 nx[U];
#pragma unroll
        for (int j = 0; j < U; ++j)                        // rows past the segment re-load its last needed row
            load(nx[j], y + 1 + j <= y1 ? y + 1 + j : y1);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (y + j < y1) store_out(dimg + (int64_t)(y + j) * g.d_row, b0, g.L, conv3_row<OP, C>(a, b, nx[j]));
            a = b;
            b = nx[j];
        }
    }
}

template <int OP, int C>
__global__ __launch_bounds__(kThreads) void conv3_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                         ConvGeom g)
{
    unsigned img; int y0, y1, sb0;
    if (!wave_item(g, img, y0, y1, sb0)) return;
    const int b0 = sb0 + (int)(threadIdx.x & 63u) * 16;
    if (b0 >= g.L) return;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    constexpr int HB = 4 * Win<C, 1>::HD;
    if (sb0 >= HB && sb0 + kStripBytes + HB <= g.L) conv3_walk<OP, C, false>(g, simg, dimg, y0, y1, b0);
    else conv3_walk<OP, C, true>(g, simg, dimg, y0, y1, b0);
}

// ---- separable fixed-point filter, K <= 7, register sliding window ---------------------------------------
template <int K> struct Taps { uint32_t kx[K]; uint32_t ky[K]; };

template <int C, int K>
__device__ __forceinline__ void hpass(uint32_t (&h)[16], const Win<C, K / 2>& W, const Taps<K>& t)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) acc = __umul24(t.kx[j], W.at(i + C * (j - K / 2))) + acc;
        h[i] = acc;                                           // Q8.8, <= 255*256
    }
}

template <int C, int K, bool EDGE>
__device__ __forceinline__ void sep_walk(const ConvGeom& g, const Taps<K>& t, const uint8_t* simg, uint8_t* dimg,
                                         int y0, int y1, int b0)
{
    constexpr int R = K / 2;
    typedef Win<C, R> WR;
    constexpr int U = EDGE ? 1 : (K <= 3 ? 3 : 2);         // rows loaded ahead per step; must divide the K-unroll
    const bool interior = !EDGE || (b0 >= 4 * WR::HD && b0 + 16 + 4 * WR::HD <= g.L);
    Gather<C, R> G;
    if (EDGE && !interior) G.init(b0, g.w);
    auto load = [&](WR& W, int y) {
        const uint8_t* rp = row_ptr<R>(g, simg, y);
        if (EDGE) load_win_edge<C, R>(W, rp, b0, interior, G);
        else load_win_fast<C, R>(W, rp, b0);
    };

    uint32_t hw[K][16];                                      // ring of K row-pass results (statically indexed)
#pragma unroll
    for (int s = 0; s < K - 1; ++s) {                        // prologue: rows y0-R .. y0+R-1
        WR W;
        load(W, y0 - R + s);
        hpass<C, K>(hw[s], W, t);
    }
    for (int y = y0; y < y1; y += K) {
#pragma unroll
        for (int s0 = 0; s0 < K; s0 += U) {
            WR nx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = s0 + u;
                if (s < K) load(nx[u], y + s < y1 ? y + s + R : y1 - 1 + R);   // clamp: rows nobody needs are not read
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = s0 + u;
                if (s >= K) continue;
                hpass<C, K>(hw[(K - 1 + s) % K], nx[u], t);
                if (y + s < y1) {
                    u32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        uint32_t word = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            uint32_t acc = 32768u;
#pragma unroll
                            for (int i = 0; i < K; ++i) acc = __umul24(t.ky[i], hw[(s + i) % K][4 * k + j]) + acc;
                            acc >>= 16;                       // taps sum to 256 -> acc <= 255
                            word |= (acc > 255u ? 255u : acc) << (8 * j);
                        }
                        o[k] = word;
                    }
                    store_out(dimg + (int64_t)(y + s) * g.d_row, b0, g.L, o);
                }
            }
        }
    }
}

template <int C, int K>
__global__ __launch_bounds__(kThreads) void sep_small_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                             ConvGeom g, Taps<K> t)
{
    unsigned img; int y0, y1, sb0;
    if (!wave_item(g, img, y0, y1, sb0)) return;
    const int b0 = sb0 + (int)(threadIdx.x & 63u) * 16;
    if (b0 >= g.L) return;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    constexpr int HB = 4 * Win<C, K / 2>::HD;
    if (sb0 >= HB && sb0 + kStripBytes + HB <= g.L) sep_walk<C, K, false>(g, t, simg, dimg, y0, y1, b0);
    else sep_walk<C, K, true>(g, t, simg, dimg, y0, y1, b0);
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
    const int y = trow - r;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    const uint8_t* rp;
    if (y < 0 && g.top) rp = g.top + (int64_t)(r + y) * g.s_row;
    else if (y >= g.h && g.bot) rp = g.bot + (int64_t)(y - g.h) * g.s_row;
    else rp = simg + (int64_t)reflect101(y, g.h) * g.s_row;
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

// ---- host side ------------------------------------------------------------------------------------------
int make_conv_geom(const char* fn, ConvGeom& g, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                   int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                   const uint8_t* top, const uint8_t* bot)
{
    if (int rc = ipx_check_image_args(fn, src, dst, n, h, w, c, c, s_row, s_img, d_row, d_img)) return rc;
    if ((top || bot) && n != 1) return ipx_fail(IPX_ERR_INVALID, "%s: halo rows need n == 1 (got %d)", fn, n);
    if ((int64_t)w * c > (1 << 30)) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: row of %lld bytes too long", fn, (long long)w * c);
    g.h = h; g.w = w; g.L = w * c;
    g.s_row = s_row; g.s_img = s_img; g.d_row = d_row; g.d_img = d_img;
    g.top = top; g.bot = bot;
    g.nstrips = (g.L + kStripBytes - 1) / kStripBytes;
    // rows per segment: aim at >= ~6k waves on the chip, but keep segments >= 16 rows (halo re-reads <= 2r/16)
    const int64_t waves_if_one_row = (int64_t)n * g.nstrips * h;
    int64_t seg = waves_if_one_row / 6144;
    if (seg < 16) seg = 16;
    if (seg > 256) seg = 256;
    if (seg > h) seg = h;
    g.seg_rows = (int)seg;
    g.nsegs = (h + g.seg_rows - 1) / g.seg_rows;
    const uint64_t items = (uint64_t)n * g.nsegs * g.nstrips;
    if (items > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: too many work items", fn);
    g.nitems = (unsigned)items;
    return IPX_OK;
}

template <int OP>
int launch_conv3(const char* fn, void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                 int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img, const uint8_t* top, const uint8_t* bot)
{
    ConvGeom g;
    if (int rc = make_conv_geom(fn, g, src, dst, n, h, w, c, s_row, s_img, d_row, d_img, top, bot)) return rc;
    const dim3 grid((g.nitems + kWavesPerBlock - 1) / kWavesPerBlock), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
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
    for (int i = 0; i < K; ++i) { t.kx[i] = kx[i]; t.ky[i] = ky[i]; }
    const dim3 grid((g.nitems + kWavesPerBlock - 1) / kWavesPerBlock), block(kThreads);
    hipLaunchKernelGGL((sep_small_kernel<C, K>), grid, block, 0, st, src, dst, g, t);
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

}  // namespace

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

int ipx_sep_filter_q8_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                         int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                         const uint16_t* kx_q8, const uint16_t* ky_q8, int k,
                         const uint8_t* top_halo, const uint8_t* bottom_halo)
{
    const char* fn = "ipx_sep_filter_q8_u8";
    ConvGeom g;
    if (int rc = make_conv_geom(fn, g, src, dst, n, h, w, c, s_row, s_img, d_row, d_img, top_halo, bottom_halo)) return rc;
    if (!kx_q8 || !ky_q8) return ipx_fail(IPX_ERR_INVALID, "%s: null taps", fn);
    if (k < 1 || (k & 1) == 0 || k > IPX_MAX_GAUSS_K) return ipx_fail(IPX_ERR_INVALID, "%s: k=%d must be odd and in 1..%d", fn, k, IPX_MAX_GAUSS_K);
    unsigned sx = 0, sy = 0;
    for (int i = 0; i < k; ++i) {
        if (kx_q8[i] > 256 || ky_q8[i] > 256) return ipx_fail(IPX_ERR_INVALID, "%s: tap %d exceeds 256 (Q0.8 of 1.0)", fn, i);
        sx += kx_q8[i]; sy += ky_q8[i];
    }
    if (sx != 256 || sy != 256) return ipx_fail(IPX_ERR_INVALID, "%s: taps must sum to 256 (got %u, %u)", fn, sx, sy);
    hipStream_t st = (hipStream_t)stream;
    if (k == 3) { launch_sep_small_c<3>(st, src, dst, g, c, kx_q8, ky_q8); IPX_LAUNCH_CHECK("sep_small_kernel<3>"); return IPX_OK; }
    if (k == 5) { launch_sep_small_c<5>(st, src, dst, g, c, kx_q8, ky_q8); IPX_LAUNCH_CHECK("sep_small_kernel<5>"); return IPX_OK; }
    if (k == 7) { launch_sep_small_c<7>(st, src, dst, g, c, kx_q8, ky_q8); IPX_LAUNCH_CHECK("sep_small_kernel<7>"); return IPX_OK; }

    // generic two-pass path through a Q8.8 scratch image of (h + 2r) rows per image
    if (n > 65535 || h + k > 65535) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: n or h too large for the two-pass path", fn);
    const int r = k / 2;
    const size_t tmp_bytes = (size_t)n * (size_t)(h + 2 * r) * (size_t)g.L * sizeof(uint16_t);
    uint16_t* tmp = nullptr;
    IPX_HIP(hipMallocAsync((void**)&tmp, tmp_bytes, st));
    TapsBig tx{}, ty{};
    for (int i = 0; i < k; ++i) { tx.k[i] = kx_q8[i]; ty.k[i] = ky_q8[i]; }
    unsigned gx = (unsigned)((g.L + 255) / 256);
    if (gx > 64) gx = 64;
    int rc = IPX_OK;
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

int ipx_gaussian_blur_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                         int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img, int k,
                         const uint8_t* top_halo, const uint8_t* bottom_halo)
{
    std::vector<uint16_t> q(k > 0 ? k : 1);
    if (int rc = ipx_gauss_kernel_q8(k, 0.0, q.data())) return rc;
    return ipx_sep_filter_q8_u8(stream, src, dst, n, h, w, c, s_row, s_img, d_row, d_img, q.data(), q.data(), k,
                                top_halo, bottom_halo);
}

}  // extern "C"

// ipx_rank.hip -- rank and morphological filters of the reference's `median` pipeline (server.py:390-441) and of
// mode=denoise_manual with method=median (server.py:214-216).  SURVEY.md section 8(f) row 4.  Gray images.
//
//   cv2.medianBlur(img, k)        uint8, odd k (3..31 here), BORDER_REPLICATE: the (k*k-1)/2-th smallest of the k x k
//                                 window.  Exact integer result, so any algorithm agrees: k = 3, 5, 7 run a selection
//                                 network in registers (two pixels per thread, packed 16-bit min/max); larger k an 8-step
//                                 binary search on the value, counting window pixels <= mid from an LDS tile.
//   cv2.erode / cv2.dilate        min / max over the structuring element's non-zero positions (anchor = centre); pixels
//                                 outside the image are ignored (OpenCV's default border value: +inf for erode, -inf for
//                                 dilate).  cv2.morphologyEx OPEN = dilate(erode), CLOSE = erode(dilate): composed by the
//                                 host from these two.
// OpenCV is absent from this image: these restate its published definitions (parity unpinned).
#include "ipx_common.h"

namespace {

constexpr int kTileW = 32, kTileH = 8, kMaxR = 15;
constexpr int kPitch = kTileW + 2 * kMaxR + 2;             // 64

struct RankGeom {
    int h, w, rx, ry;                                      // window radii
    int64_t s_row, s_img, d_row, d_img;
    int tiles_x, tiles_y;
};
struct SeMask { uint32_t bits[31]; };                      // row i, bit j: structuring element (i, j) is set

template <int FILL>   // -1: replicate the border; else the byte to use outside the image
__device__ __forceinline__ void load_tile(uint8_t* s_tile, const uint8_t* simg, const RankGeom& g, int tx0, int ty0, int tid)
{
    const int tw = kTileW + 2 * g.rx, th = kTileH + 2 * g.ry;
    for (int i = tid; i < tw * th; i += kTileW * kTileH) {
        const int ry = i / tw, rx = i - ry * tw;
        int y = ty0 - g.ry + ry, x = tx0 - g.rx + rx;
        const bool out = y < 0 || y >= g.h || x < 0 || x >= g.w;
        y = y < 0 ? 0 : y >= g.h ? g.h - 1 : y;
        x = x < 0 ? 0 : x >= g.w ? g.w - 1 : x;
        const uint8_t v = simg[(int64_t)y * g.s_row + x];
        s_tile[ry * kPitch + rx] = (FILL >= 0 && out) ? (uint8_t)FILL : v;
    }
}

__global__ __launch_bounds__(kTileW * kTileH) void median_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, RankGeom g)
{
    __shared__ uint8_t s_tile[(kTileH + 2 * kMaxR) * kPitch];
    const unsigned tile = blockIdx.x;
    const int tx0 = (int)(tile % (unsigned)g.tiles_x) * kTileW;
    const int ty0 = (int)(tile / (unsigned)g.tiles_x % (unsigned)g.tiles_y) * kTileH;
    const unsigned img = tile / ((unsigned)g.tiles_x * (unsigned)g.tiles_y);
    const int tid = (int)threadIdx.x;
    load_tile<-1>(s_tile, src + (int64_t)img * g.s_img, g, tx0, ty0, tid);
    __syncthreads();
    const int lx = tid % kTileW, ly = tid / kTileW;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= g.w || y >= g.h) return;
    const int K = 2 * g.rx + 1;
    const int need = (K * K) / 2 + 1;                      // the median is the smallest v with count(<= v) >= need
    const uint8_t* base = s_tile + ly * kPitch + lx;
    int lo = 0, hi = 255;
#pragma unroll 1
    for (int step = 0; step < 8; ++step) {
        const int mid = (lo + hi) >> 1;
        int cnt = 0;
        for (int wy = 0; wy < K; ++wy) {
            const uint8_t* rp = base + wy * kPitch;
            for (int wx = 0; wx < K; ++wx) cnt += rp[wx] <= mid ? 1 : 0;
        }
        if (cnt >= need) hi = mid; else lo = mid + 1;
    }
    dst[(int64_t)img * g.d_img + (int64_t)y * g.d_row + x] = (uint8_t)lo;
}

// ---- median for k = 3, 5, 7: forgetful selection in registers ----------------------------------------------------
// A thread owns two horizontally adjacent pixels and keeps their window values as packed 16-bit pairs, so every
// compare-exchange is one v_pk_min_u16 + one v_pk_max_u16 for both pixels.  Forgetful selection: the median of n values
// can never be the minimum or the maximum of any n/2 + 2 of them, so start with n/2 + 2 values, drop their min and max,
// take in the next value, and repeat until three are left; their middle one is the median.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void cswap(us2& a, us2& b)             // a <- min, b <- max (per 16-bit half)
{
    const us2 lo = __builtin_elementwise_min(a, b), hi = __builtin_elementwise_max(a, b);
    a = lo;
    b = hi;
}

template <int CNT>
__device__ __forceinline__ void min_max_to_ends(us2 (&v)[CNT])   // v[0] <- min, v[CNT-1] <- max of v[0..CNT)
{
#pragma unroll
    for (int i = 0; i < CNT / 2; ++i) cswap(v[i], v[CNT - 1 - i]);
#pragma unroll
    for (int i = 1; i < (CNT + 1) / 2; ++i) cswap(v[0], v[i]);
#pragma unroll
    for (int i = CNT / 2; i < CNT - 1; ++i) cswap(v[i], v[CNT - 1]);
}

template <int CNT, int N, typename NextFn>
struct Forget {
    static __device__ __forceinline__ us2 run(us2 (&v)[CNT], int next, NextFn& fetch)
    {
        min_max_to_ends<CNT>(v);
        us2 w[CNT - 1];
        w[0] = fetch(next);                                       // replaces the minimum; the maximum falls off the end
#pragma unroll
        for (int i = 1; i < CNT - 1; ++i) w[i] = v[i];
        return Forget<CNT - 1, N, NextFn>::run(w, next + 1, fetch);
    }
};
template <int N, typename NextFn>
struct Forget<3, N, NextFn> {
    static __device__ __forceinline__ us2 run(us2 (&v)[3], int, NextFn&)
    {
        min_max_to_ends<3>(v);
        return v[1];
    }
};

constexpr int kSmW = 64, kSmH = 8;                               // 64 x 8 pixel tile, 2 pixels per thread
constexpr int kSmPitch = kSmW + 2 * 3 + 2;                       // 72

template <int K>
__global__ __launch_bounds__(256) void median_small_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, RankGeom g)
{
    constexpr int R = K / 2, N = K * K, M = N / 2 + 2;
    __shared__ uint8_t s_tile[(kSmH + 2 * R) * kSmPitch];
    const unsigned tile = blockIdx.x;
    const int tx0 = (int)(tile % (unsigned)g.tiles_x) * kSmW;
    const int ty0 = (int)(tile / (unsigned)g.tiles_x % (unsigned)g.tiles_y) * kSmH;
    const unsigned img = tile / ((unsigned)g.tiles_x * (unsigned)g.tiles_y);
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    const int tid = (int)threadIdx.x;
    constexpr int tw = kSmW + 2 * R, th = kSmH + 2 * R;
    for (int i = tid; i < tw * th; i += 256) {                   // BORDER_REPLICATE: clamp the coordinates
        const int ry = i / tw, rx = i - ry * tw;
        int y = ty0 - R + ry, x = tx0 - R + rx;
        y = y < 0 ? 0 : y >= g.h ? g.h - 1 : y;
        x = x < 0 ? 0 : x >= g.w ? g.w - 1 : x;
        s_tile[ry * kSmPitch + rx] = simg[(int64_t)y * g.s_row + x];
    }
    __syncthreads();
    const int lx = 2 * (tid % (kSmW / 2)), ly = tid / (kSmW / 2);
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= g.w || y >= g.h) return;
    const uint8_t* base = s_tile + ly * kSmPitch + lx;
    auto fetch = [&](int i) {                                     // window value i of pixel x (low half) and x+1 (high half)
        const uint8_t* p = base + (i / K) * kSmPitch + (i % K);
        us2 r;
        r.x = p[0];
        r.y = p[1];
        return r;
    };
    us2 v[M];
#pragma unroll
    for (int i = 0; i < M; ++i) v[i] = fetch(i);
    const us2 med = Forget<M, N, decltype(fetch)>::run(v, M, fetch);
    uint8_t* dp = dst + (int64_t)img * g.d_img + (int64_t)y * g.d_row + x;
    dp[0] = (uint8_t)med.x;
    if (x + 1 < g.w) dp[1] = (uint8_t)med.y;
}

template <bool DILATE>
__global__ __launch_bounds__(kTileW * kTileH) void morph_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, RankGeom g, SeMask se)
{
    __shared__ uint8_t s_tile[(kTileH + 2 * kMaxR) * kPitch];
    const unsigned tile = blockIdx.x;
    const int tx0 = (int)(tile % (unsigned)g.tiles_x) * kTileW;
    const int ty0 = (int)(tile / (unsigned)g.tiles_x % (unsigned)g.tiles_y) * kTileH;
    const unsigned img = tile / ((unsigned)g.tiles_x * (unsigned)g.tiles_y);
    const int tid = (int)threadIdx.x;
    load_tile<DILATE ? 0 : 255>(s_tile, src + (int64_t)img * g.s_img, g, tx0, ty0, tid);
    __syncthreads();
    const int lx = tid % kTileW, ly = tid / kTileW;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= g.w || y >= g.h) return;
    const int KW = 2 * g.rx + 1, KH = 2 * g.ry + 1;
    const uint8_t* base = s_tile + ly * kPitch + lx;
    uint32_t acc = DILATE ? 0u : 255u;
    bool any = false;
    for (int wy = 0; wy < KH; ++wy) {
        const uint32_t bits = se.bits[wy];                 // block-uniform
        const uint8_t* rp = base + wy * kPitch;
        for (int wx = 0; wx < KW; ++wx) {
            if ((bits >> wx) & 1u) {
                const uint32_t v = rp[wx];
                acc = DILATE ? (v > acc ? v : acc) : (v < acc ? v : acc);
                any = true;
            }
        }
    }
    (void)any;
    dst[(int64_t)img * g.d_img + (int64_t)y * g.d_row + x] = (uint8_t)acc;
}

int make_geom(const char* fn, RankGeom& g, const uint8_t* src, uint8_t* dst, int n, int h, int w, int rx, int ry,
              int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img, unsigned* blocks)
{
    if (int rc = ipx_check_image_args(fn, src, dst, n, h, w, 1, 1, s_row, s_img, d_row, d_img)) return rc;
    g.h = h; g.w = w; g.rx = rx; g.ry = ry;
    g.s_row = s_row; g.s_img = s_img; g.d_row = d_row; g.d_img = d_img;
    g.tiles_x = (w + kTileW - 1) / kTileW;
    g.tiles_y = (h + kTileH - 1) / kTileH;
    const uint64_t nb = (uint64_t)n * g.tiles_x * g.tiles_y;
    if (nb > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: too many tiles", fn);
    *blocks = (unsigned)nb;
    return IPX_OK;
}

}  // namespace

extern "C" {

int ipx_median_blur_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                       int64_t src_row_stride, int64_t src_img_stride, int64_t dst_row_stride, int64_t dst_img_stride, int k)
{
    const char* fn = "ipx_median_blur_u8";
    ipx_clear_error();
    if (k < 1 || (k & 1) == 0 || k > 2 * kMaxR + 1)
        return ipx_fail(IPX_ERR_INVALID, "%s: k=%d must be odd and in 1..%d", fn, k, 2 * kMaxR + 1);
    if (src == dst) return ipx_fail(IPX_ERR_INVALID, "%s: in-place operation is not supported", fn);
    RankGeom g;
    unsigned blocks = 0;
    if (int rc = make_geom(fn, g, src, dst, n, h, w, k / 2, k / 2, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride, &blocks)) return rc;
    if (k == 3 || k == 5 || k == 7) {                             // register selection network, 2 pixels per thread
        g.tiles_x = (w + kSmW - 1) / kSmW;
        g.tiles_y = (h + kSmH - 1) / kSmH;
        const uint64_t nb = (uint64_t)n * g.tiles_x * g.tiles_y;
        if (nb > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: too many tiles", fn);
        const dim3 grid((unsigned)nb), block(256);
        if (k == 3) hipLaunchKernelGGL(median_small_kernel<3>, grid, block, 0, (hipStream_t)stream, src, dst, g);
        else if (k == 5) hipLaunchKernelGGL(median_small_kernel<5>, grid, block, 0, (hipStream_t)stream, src, dst, g);
        else hipLaunchKernelGGL(median_small_kernel<7>, grid, block, 0, (hipStream_t)stream, src, dst, g);
        IPX_LAUNCH_CHECK("median_small_kernel");
        return IPX_OK;
    }
    hipLaunchKernelGGL(median_kernel, dim3(blocks), dim3(kTileW * kTileH), 0, (hipStream_t)stream, src, dst, g);
    IPX_LAUNCH_CHECK("median_kernel");
    return IPX_OK;
}

int ipx_morph_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                 int64_t src_row_stride, int64_t src_img_stride, int64_t dst_row_stride, int64_t dst_img_stride,
                 int op, const uint8_t* se_host, int se_rows, int se_cols)
{
    const char* fn = "ipx_morph_u8";
    ipx_clear_error();
    if (op != 0 && op != 1) return ipx_fail(IPX_ERR_INVALID, "%s: op=%d (0 = erode, 1 = dilate)", fn, op);
    if (!se_host) return ipx_fail(IPX_ERR_INVALID, "%s: null structuring element", fn);
    if (se_rows < 1 || se_cols < 1 || !(se_rows & 1) || !(se_cols & 1) || se_rows > 2 * kMaxR + 1 || se_cols > 2 * kMaxR + 1)
        return ipx_fail(IPX_ERR_INVALID, "%s: structuring element %d x %d must have odd sides in 1..%d", fn, se_rows, se_cols, 2 * kMaxR + 1);
    if (src == dst) return ipx_fail(IPX_ERR_INVALID, "%s: in-place operation is not supported", fn);
    SeMask se{};
    bool any = false;
    for (int i = 0; i < se_rows; ++i)
        for (int j = 0; j < se_cols; ++j)
            if (se_host[i * se_cols + j]) { se.bits[i] |= 1u << j; any = true; }
    if (!any) return ipx_fail(IPX_ERR_INVALID, "%s: empty structuring element", fn);
    RankGeom g;
    unsigned blocks = 0;
    if (int rc = make_geom(fn, g, src, dst, n, h, w, se_cols / 2, se_rows / 2, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride, &blocks)) return rc;
    if (op) hipLaunchKernelGGL(morph_kernel<true>, dim3(blocks), dim3(kTileW * kTileH), 0, (hipStream_t)stream, src, dst, g, se);
    else hipLaunchKernelGGL(morph_kernel<false>, dim3(blocks), dim3(kTileW * kTileH), 0, (hipStream_t)stream, src, dst, g, se);
    IPX_LAUNCH_CHECK("morph_kernel");
    return IPX_OK;
}

}  // extern "C"

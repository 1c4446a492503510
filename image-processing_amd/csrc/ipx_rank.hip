// ipx_rank.hip -- rank and morphological filters of the reference's `median` pipeline (server.py:390-441) and of
// mode=denoise_manual with method=median (server.py:214-216).  SURVEY.md section 8(f) row 4.  Gray images.
//
//   cv2.medianBlur(img, k)        uint8, odd k (3..31 here), BORDER_REPLICATE: the (k*k-1)/2-th smallest of the k x k
//                                 window.  Exact integer result, so any algorithm agrees: k <= 9 run a forgetful
//                                 selection in registers (two pixels per thread, packed 16-bit min/max); larger k
//                                 sliding per-column histograms in LDS (constant work per pixel).
//   cv2.erode / cv2.dilate        min / max over the structuring element's non-zero positions (anchor = centre); pixels
//                                 outside the image are ignored (OpenCV's default border value: +inf for erode, -inf for
//                                 dilate).  cv2.morphologyEx OPEN = dilate(erode), CLOSE = erode(dilate): composed by the
//                                 host from these two.
// OpenCV is absent from this image: these restate its published definitions (parity unpinned).
#include "ipx_common.h"
#include <stdlib.h>

int ipx_median3_walker(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                       int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img);
int ipx_median5_walker(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                       int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img);
int ipx_median7_walker(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                       int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img);
int ipx_median9_walker(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                       int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img);
int ipx_median11_walker(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                        int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img);
int ipx_morph_small(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                    int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img, int dilate,
                    const uint32_t* row_bits, int se_rows, int se_cols);
int ipx_median_rows(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                    int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img, int k);

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

// ---- median for k >= 11: sliding histograms (constant work per pixel, whatever k) --------------------------------
// One wave owns a strip of 64 output columns and walks down a band of rows.  LDS holds one 256-bin byte histogram per
// strip column (64 + 2r of them), each over the 2r+1 rows around the current one; stepping down a row takes one pixel
// out of and puts one into every column histogram (a lane per column).  Along a row the window histogram lives in
// registers, four 16-bit bins per lane: moving right adds the entering column's histogram and subtracts the leaving
// one's (one conflict-free dword read each), then a wave prefix sum over the lane totals finds the lane and, inside
// it, the bin where the running count first reaches (k*k)/2 + 1.
// Column c's dwords are rotated by c, so that lanes bumping the same value in 64 different columns hit 64 banks.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
constexpr int kHsW = 64;

__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8   -> scan inside each row of 16
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ us2 bins_lo(uint32_t d) { return __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, d, 0x0c010c00u)); }
__device__ __forceinline__ us2 bins_hi(uint32_t d) { return __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, d, 0x0c030c02u)); }

// The wave's window histogram (4 bins per lane) -> the bin where the running count first reaches `need`, written to
// s_out[slot] by the one lane that holds it: that lane's inclusive prefix has reached `need` and its exclusive prefix
// has not (lane 63's inclusive prefix is k*k, so there always is one).  Nothing goes through scalar registers
// (ballot / s_ff1 / v_readlane made every output pixel a vector -> scalar -> vector round trip: 590 cycles per pixel and
// wave where the instructions need 160), so the searches of consecutive pixels overlap.
// Two pixels per search: the lane totals of both window histograms share one register (16 bits each: k*k <= 961, so the
// low half never carries into the high one) and ONE wave prefix sum serves both.
__device__ __forceinline__ void find_rank_store2(us2 a01, us2 a23, us2 b01, us2 b23, int need, int lane, uint8_t* s_out, int slot_a, int slot_b)
{
    const us2 ones = {1, 1};
    const uint32_t ta = __builtin_amdgcn_udot2(a01, ones, __builtin_amdgcn_udot2(a23, ones, 0u, false), false);
    const uint32_t tb = __builtin_amdgcn_udot2(b01, ones, __builtin_amdgcn_udot2(b23, ones, 0u, false), false);
    const uint32_t p = (uint32_t)wave_inclusive_scan((int)(ta | (tb << 16)));
    // every lane stores -- the others into bytes of their own behind the results -- so that there is no branch
    auto resolve = [&](uint32_t pp, uint32_t t, us2 h01, us2 h23, int slot, int spare) {
        const int below = (int)(pp - t);
        const uint32_t a = __builtin_bit_cast(uint32_t, h01), b = __builtin_bit_cast(uint32_t, h23);
        const int c0 = below + (int)(a & 0xffffu), c1 = c0 + (int)(a >> 16), c2 = c1 + (int)(b & 0xffffu);
        s_out[(int)pp >= need && below < need ? slot : spare] = (uint8_t)(4 * lane + (c0 < need) + (c1 < need) + (c2 < need));
    };
    resolve(p & 0xffffu, ta, a01, a23, slot_a, kHsW + lane);
    resolve(p >> 16, tb, b01, b23, slot_b, 2 * kHsW + lane);
}

__global__ __launch_bounds__(64) void median_hist_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, RankGeom g, int band_rows)
{
    extern __shared__ uint32_t s_hist[];                            // (64 + 2r) columns x 256 byte bins
    uint8_t* hb = (uint8_t*)s_hist;
    const unsigned tile = blockIdx.x;
    const int x0 = (int)(tile % (unsigned)g.tiles_x) * kHsW;
    const int y0 = (int)(tile / (unsigned)g.tiles_x % (unsigned)g.tiles_y) * band_rows;
    const unsigned img = tile / ((unsigned)g.tiles_x * (unsigned)g.tiles_y);
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    const int lane = (int)threadIdx.x;
    const int r = g.rx, K = 2 * r + 1, nc = kHsW + 2 * r, need = (K * K) / 2 + 1;
    uint8_t* s_out = hb + nc * 256;                                 // the row's 64 results, then 128 bytes nobody reads
    const int y1 = y0 + band_rows < g.h ? y0 + band_rows : g.h;
    const int nout = g.w - x0 < kHsW ? g.w - x0 : kHsW;

    // the (up to) two columns whose histograms this lane maintains
    int xc[2];
    bool own[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = lane + 64 * q;
        own[q] = c < nc;
        const int x = x0 - r + c;
        xc[q] = x < 0 ? 0 : x >= g.w ? g.w - 1 : x;                   // BORDER_REPLICATE
    }
    auto bin_addr = [&](int q, int v) { const int c = lane + 64 * q; return c * 256 + ((((v >> 2) + c) & 63) << 2) + (v & 3); };
    auto row_of = [&](int y) { return (int64_t)(y < 0 ? 0 : y >= g.h ? g.h - 1 : y) * g.s_row; };

    for (int i = lane; i < nc * 64; i += 64) s_hist[i] = 0u;
    __syncthreads();
    for (int j = -r; j <= r; ++j) {
        const int64_t ro = row_of(y0 + j);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int v = simg[ro + xc[q]];
            if (own[q]) hb[bin_addr(q, v)] += 1;
        }
    }
    __syncthreads();

    for (int y = y0; y < y1; ++y) {
        // the pixels that leave and enter the column histograms after this row (loaded early, used after the row)
        int ov[2], nv[2];
        {
            const int64_t ro = row_of(y - r), rn = row_of(y + r + 1);
#pragma unroll
            for (int q = 0; q < 2; ++q) { ov[q] = simg[ro + xc[q]]; nv[q] = simg[rn + xc[q]]; }
        }
        // two streams per wave, output columns 0 .. 31 and 32 .. 63, in lockstep: their instructions interleave (one
        // stream alone left the wave stalled for half of its 390 cycles per pixel) and share the prefix sum
        constexpr int HW = kHsW / 2;
        us2 a01 = {0, 0}, a23 = {0, 0}, b01 = {0, 0}, b23 = {0, 0};
        auto column = [&](int c) { c = c < nc ? c : nc - 1; return s_hist[c * 64 + ((lane + c) & 63)]; };
        for (int c = 0; c < K; ++c) {
            const uint32_t da = column(c), db = column(HW + c);
            a01 += bins_lo(da);
            a23 += bins_hi(da);
            b01 += bins_lo(db);
            b23 += bins_hi(db);
        }
        find_rank_store2(a01, a23, b01, b23, need, lane, s_out, 0, HW);
        uint32_t an = column(1 + 2 * r), ao = column(0), bn = column(HW + 1 + 2 * r), bo = column(HW);
        // always 32 steps (the strip holds 64 + 2r column histograms whatever the image width -- columns past its right
        // edge are replicas of the last one -- and results past the edge are not stored): a fixed trip count unrolls
#pragma unroll 4
        for (int i = 1; i < HW; ++i) {
            const uint32_t an2 = column(i + 1 + 2 * r), ao2 = column(i);          // next step's, in flight during the search
            const uint32_t bn2 = column(HW + i + 1 + 2 * r), bo2 = column(HW + i);
            a01 = a01 + bins_lo(an) - bins_lo(ao);
            a23 = a23 + bins_hi(an) - bins_hi(ao);
            b01 = b01 + bins_lo(bn) - bins_lo(bo);
            b23 = b23 + bins_hi(bn) - bins_hi(bo);
            find_rank_store2(a01, a23, b01, b23, need, lane, s_out, i, HW + i);
            an = an2; ao = ao2; bn = bn2; bo = bo2;
        }
        if (lane < nout) dimg[(int64_t)y * g.d_row + x0 + lane] = s_out[lane];   // (a wave's LDS accesses complete in order)
        __syncthreads();
        if (y + 1 < y1) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (own[q]) {
                    hb[bin_addr(q, ov[q])] -= 1;
                    hb[bin_addr(q, nv[q])] += 1;
                }
        }
        __syncthreads();
    }
}

// ---- median for k = 3, 5, 7, 9: forgetful selection in registers ----------------------------------------------------
// A thread owns two horizontally adjacent pixels and keeps their window values as packed 16-bit pairs, so every
// compare-exchange is one v_pk_min_u16 + one v_pk_max_u16 for both pixels.  Forgetful selection: the median of n values
// can never be the minimum or the maximum of any n/2 + 2 of them, so start with n/2 + 2 values, drop their min and max,
// take in the next value, and repeat until three are left; their middle one is the median.
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
constexpr int kSmPitch = kSmW + 2 * 4 + 4;                       // 76

template <int K>
__global__ __launch_bounds__(256) void median_small_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, RankGeom g)
{
    constexpr int R = K / 2, N = K * K, M = K == 1 ? 3 : N / 2 + 2;
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
    us2 med;
    if constexpr (K == 1) {
        med = fetch(0);
    } else {
        us2 v[M];
#pragma unroll
        for (int i = 0; i < M; ++i) v[i] = fetch(i);
        med = Forget<M, N, decltype(fetch)>::run(v, M, fetch);
    }
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
    const int KH = 2 * g.ry + 1;
    const uint8_t* base = s_tile + ly * kPitch + lx;
    uint32_t acc = DILATE ? 0u : 255u;
    for (int wy = 0; wy < KH; ++wy) {
        uint32_t bits = se.bits[wy];                       // block-uniform: walk the set positions only
        const uint8_t* rp = base + wy * kPitch;
        while (bits) {
            const uint32_t v = rp[__builtin_ctz(bits)];
            bits &= bits - 1;
            acc = DILATE ? (v > acc ? v : acc) : (v < acc ? v : acc);
        }
    }
    dst[(int64_t)img * g.d_img + (int64_t)y * g.d_row + x] = (uint8_t)acc;
}

// ---- erode / dilate with a structuring element larger than 5 x 5: runs and a sparse table ---------------------------------
// The kernel above visits every member of the element for every pixel: 729 LDS reads for OpenCV's 31 x 31 ellipse.  A row of a
// structuring element is a few RUNS of members (one per row for every ellipse, rectangle and cross), and the min / max over
// a run of length L is op(T_p[s], T_p[s + L - p]), p = the largest power of two <= L, where T_p[x] = op over tile[x .. x + p - 1]
// (the sparse table of range-minimum queries, built by doubling: T_2p[x] = op(T_p[x], T_p[x + p])).  The block stages its tile
// (pixels outside the image = the neutral value, OpenCV's default morphology border), builds the levels the element needs
// in LDS, and a pixel then costs two byte reads per run: 62 for the 31 x 31 ellipse.  A thread takes 8 pixels of one
// column (rows 4 apart): a run's two LDS offsets are formed once and reach the 8 rows through immediate offsets.
constexpr int kRunW = 64, kRunH = 32, kRunThreads = 256, kRunPx = kRunW * kRunH / kRunThreads;     // 8 pixels per thread
constexpr int kRunPitch = kRunW + 2 * kMaxR + 18;          // 112: the levels read up to 15 bytes past a tile row
constexpr int kMaxRuns = 64;
struct SeRuns {
    int count, levels, plane;                              // runs; table levels beyond the tile itself (0 .. 4); LDS bytes per level
    uint16_t off1[kMaxRuns], off2[kMaxRuns];               // LDS byte offsets of the two queries, relative to the pixel's tile origin
};

template <bool DILATE>
__global__ __launch_bounds__(kRunThreads) void morph_runs_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, RankGeom g, SeRuns se)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_t[];   // (levels + 1) planes of th x kRunPitch + 16 bytes
    const unsigned tile = blockIdx.x;
    const int tx0 = (int)(tile % (unsigned)g.tiles_x) * kRunW;
    const int ty0 = (int)(tile / (unsigned)g.tiles_x % (unsigned)g.tiles_y) * kRunH;
    const unsigned img = tile / ((unsigned)g.tiles_x * (unsigned)g.tiles_y);
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    const int tid = (int)threadIdx.x;
    constexpr uint8_t kNeutral = DILATE ? 0 : 255;
    auto op = [](uint32_t a, uint32_t b) { return DILATE ? (a > b ? a : b) : (a < b ? a : b); };

    // Staging and the table levels work on dwords: 32 lanes per tile row (28 dwords = 112 bytes = the pitch), 8 rows per pass.
    const int tw = kRunW + 2 * g.rx, th = kRunH + 2 * g.ry;
    const int dq = tid & 31, r8 = tid >> 5;
    const int cx = 4 * dq, gx = tx0 - g.rx + cx;                                 // tile column / image column of this lane's dword
    if (dq < kRunPitch / 4) {
#pragma unroll
        for (int it = 0; it < (kRunH + 2 * kMaxR + 7) / 8; ++it) {               // unrolled: the passes' loads are issued together
            const int ry = r8 + 8 * it;
            if (ry >= th) break;
            const int y = ty0 - g.ry + ry;
            uint32_t v = kNeutral * 0x01010101u;
            if (y >= 0 && y < g.h) {
                const uint8_t* rp = simg + (int64_t)y * g.s_row;
                if (gx >= 0 && gx + 4 <= g.w && cx + 4 <= tw) {
                    v = *(const u32_u*)(rp + gx);                                // (unaligned dword loads are fine in global memory)
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int x = gx + j;
                        if (x >= 0 && x < g.w && cx + j < tw) v = (v & ~(255u << (8 * j))) | ((uint32_t)rp[x] << (8 * j));
                    }
                }
            }
            *(uint32_t*)(s_t + ry * kRunPitch + cx) = v;
        }
    }
    __syncthreads();
    // T_(2^l)[x] = op(T_(2^(l-1))[x], T_(2^(l-1))[x + 2^(l-1)]), four entries per lane: the second operand is the dword 1, 2, 4 or 8
    // bytes on (two aligned reads and a byte alignment for 1 and 2), op on bytes as two packed 16-bit operations.  Entries near a
    // row's end combine bytes of the next row: they are never queried (a query's range lies inside the tile row).
    auto op4 = [](uint32_t a, uint32_t b) {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        const uint32_t m = 0x00ff00ffu;
        const uint32_t ae = a & m, ao = (a >> 8) & m, be = b & m, bo = (b >> 8) & m;
        us2 e, o;
        const us2 ae2 = __builtin_bit_cast(us2, ae), be2 = __builtin_bit_cast(us2, be);
        const us2 ao2 = __builtin_bit_cast(us2, ao), bo2 = __builtin_bit_cast(us2, bo);
        e = DILATE ? __builtin_elementwise_max(ae2, be2) : __builtin_elementwise_min(ae2, be2);
        o = DILATE ? __builtin_elementwise_max(ao2, bo2) : __builtin_elementwise_min(ao2, bo2);
        return __builtin_bit_cast(uint32_t, e) | (__builtin_bit_cast(uint32_t, o) << 8);
    };
    for (int l = 1; l <= se.levels; ++l) {
        const uint8_t* lo = s_t + (l - 1) * se.plane;
        uint8_t* hi = s_t + l * se.plane;
        const int step = 1 << (l - 1);
        if (dq < kRunPitch / 4) {
            for (int ry = r8; ry < th; ry += 8) {
                const int a = ry * kRunPitch + cx;
                const uint32_t v0 = *(const uint32_t*)(lo + a);
                uint32_t v1;
                if (step >= 4) {
                    v1 = *(const uint32_t*)(lo + a + step);
                } else {
                    const uint32_t nx = *(const uint32_t*)(lo + a + 4);
                    v1 = __builtin_amdgcn_alignbyte(nx, v0, (uint32_t)step);
                }
                *(uint32_t*)(hi + a) = op4(v0, v1);
            }
        }
        __syncthreads();
    }

    const int lx = tid % kRunW, ly = tid / kRunW;                                // rows ly, ly + 4, ..., ly + 28
    const int x = tx0 + lx;
    if (x >= g.w) return;
    const uint8_t* base = s_t + ly * kRunPitch + lx;
    uint32_t acc[kRunPx];
#pragma unroll
    for (int i = 0; i < kRunPx; ++i) acc[i] = kNeutral;
    for (int r = 0; r < se.count; ++r) {
        const uint8_t* p1 = base + se.off1[r];
        const uint8_t* p2 = base + se.off2[r];
#pragma unroll
        for (int i = 0; i < kRunPx; ++i) acc[i] = op(acc[i], op(p1[i * 4 * kRunPitch], p2[i * 4 * kRunPitch]));
    }
    uint8_t* dp = dst + (int64_t)img * g.d_img + x;
#pragma unroll
    for (int i = 0; i < kRunPx; ++i) {
        const int y = ty0 + ly + 4 * i;
        if (y < g.h) dp[(int64_t)y * g.d_row] = (uint8_t)acc[i];
    }
}

// The element's rows as runs; false if it has more than kMaxRuns of them (the kernel above this one then takes it).
bool make_runs(const SeMask& m, int se_rows, int se_cols, SeRuns* out)
{
    SeRuns r{};
    int maxlen = 1;
    r.plane = (kRunH + se_rows - 1) * kRunPitch + 16;
    for (int wy = 0; wy < se_rows; ++wy) {
        int wx = 0;
        while (wx < se_cols) {
            if (!((m.bits[wy] >> wx) & 1u)) { ++wx; continue; }
            int len = 1;
            while (wx + len < se_cols && ((m.bits[wy] >> (wx + len)) & 1u)) ++len;
            if (r.count == kMaxRuns) return false;
            int lvl = 0;
            while ((2 << lvl) <= len) ++lvl;                                     // 2^lvl <= len < 2^(lvl + 1)
            const int o1 = lvl * r.plane + wy * kRunPitch + wx;
            r.off1[r.count] = (uint16_t)o1;
            r.off2[r.count] = (uint16_t)(o1 + len - (1 << lvl));
            ++r.count;
            if (len > maxlen) maxlen = len;
            wx += len;
        }
    }
    while ((2 << r.levels) <= maxlen) ++r.levels;
    *out = r;
    return true;
}

int make_geom(const char* fn, RankGeom& g, const uint8_t* src, uint8_t* dst, int n, int h, int w, int rx, int ry,
              int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img, unsigned* blocks)
{
    if (int rc = ipx_check_image_args(fn, src, dst, n, h, w, 1, 1, s_row, s_img, d_row, d_img)) return rc;
    if (rx > 0 || ry > 0)
        if (int rc = ipx_check_disjoint(fn, "source", src, w, s_row, s_img, "destination", dst, w, d_row, d_img, n, h)) return rc;
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
    if (k == 3) {                                                 // strip walker with sorted columns (ipx_conv.hip)
        const int rc = ipx_median3_walker((hipStream_t)stream, src, dst, n, h, w, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride);
        if (rc != 0) return rc < 0 ? rc : IPX_OK;
    }
    if (k == 5) {                                                 // the same with shared pair merges
        const int rc = ipx_median5_walker((hipStream_t)stream, src, dst, n, h, w, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride);
        if (rc != 0) return rc < 0 ? rc : IPX_OK;
    }
    if (k == 7) {
        const int rc = ipx_median7_walker((hipStream_t)stream, src, dst, n, h, w, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride);
        if (rc != 0) return rc < 0 ? rc : IPX_OK;
    }
    if (k == 9) {
        const int rc = ipx_median9_walker((hipStream_t)stream, src, dst, n, h, w, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride);
        if (rc != 0) return rc < 0 ? rc : IPX_OK;
    }
    if (k == 11) {
        const int rc = ipx_median11_walker((hipStream_t)stream, src, dst, n, h, w, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride);
        if (rc != 0) return rc < 0 ? rc : IPX_OK;
    }
    if (k <= 9) {                                                 // register selection network, 2 pixels per thread
        g.tiles_x = (w + kSmW - 1) / kSmW;
        g.tiles_y = (h + kSmH - 1) / kSmH;
        const uint64_t nb = (uint64_t)n * g.tiles_x * g.tiles_y;
        if (nb > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: too many tiles", fn);
        const dim3 grid((unsigned)nb), block(256);
        switch (k) {
        case 1: hipLaunchKernelGGL(median_small_kernel<1>, grid, block, 0, (hipStream_t)stream, src, dst, g); break;
        case 3: hipLaunchKernelGGL(median_small_kernel<3>, grid, block, 0, (hipStream_t)stream, src, dst, g); break;
        case 5: hipLaunchKernelGGL(median_small_kernel<5>, grid, block, 0, (hipStream_t)stream, src, dst, g); break;
        case 7: hipLaunchKernelGGL(median_small_kernel<7>, grid, block, 0, (hipStream_t)stream, src, dst, g); break;
        default: hipLaunchKernelGGL(median_small_kernel<9>, grid, block, 0, (hipStream_t)stream, src, dst, g); break;
        }
        IPX_LAUNCH_CHECK("median_small_kernel");
        return IPX_OK;
    }
    if (k >= 13
#ifdef IPX_DEV
        && !getenv("IPX_MEDIAN_COLHIST")                              // A/B switch of development builds
#endif
    ) {                                                               // a lane per row, per-lane window histograms (ipx_medianrow.hip)
        const int rc = ipx_median_rows((hipStream_t)stream, src, dst, n, h, w, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride, k);
        return rc < 0 ? rc : IPX_OK;
    }
    // sliding histograms: a wave per 64-column strip and row band; bands sized to give the chip a few thousand waves,
    // but never much shorter than the k rows it costs to set a band's histograms up
    const int strips = (w + kHsW - 1) / kHsW;
    int bands = (int)((4096 + (int64_t)strips * n - 1) / ((int64_t)strips * n));
    const int min_rows = k > 16 ? k : 16;
    if (bands > (h + min_rows - 1) / min_rows) bands = (h + min_rows - 1) / min_rows;
    if (bands < 1) bands = 1;
    const int band_rows = (h + bands - 1) / bands;
    g.tiles_x = strips;
    g.tiles_y = (h + band_rows - 1) / band_rows;
    const uint64_t nb = (uint64_t)n * g.tiles_x * g.tiles_y;
    if (nb > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: too many tiles", fn);
    hipLaunchKernelGGL(median_hist_kernel, dim3((unsigned)nb), dim3(64), (size_t)(kHsW + k - 1) * 256 + 3 * kHsW, (hipStream_t)stream, src, dst, g, band_rows);
    IPX_LAUNCH_CHECK("median_hist_kernel");
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
    // 3 x 3 and 5 x 5 elements with at most two distinct row masks (every rectangle, cross and ellipse): strip walker
    if (int rc = ipx_morph_small((hipStream_t)stream, src, dst, n, h, w, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride,
                                 op, se.bits, se_rows, se_cols))
        return rc < 0 ? rc : IPX_OK;
    SeRuns runs;
    if (make_runs(se, se_rows, se_cols, &runs)) {                 // rows of few runs (every ellipse, rectangle, cross): sparse-table queries
        g.tiles_x = (w + kRunW - 1) / kRunW;
        g.tiles_y = (h + kRunH - 1) / kRunH;
        const uint64_t nb = (uint64_t)n * g.tiles_x * g.tiles_y;
        if (nb > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: too many tiles", fn);
        const size_t lds = (size_t)(runs.levels + 1) * runs.plane;              // <= 5 x 6960 bytes
        if (op) hipLaunchKernelGGL(morph_runs_kernel<true>, dim3((unsigned)nb), dim3(kRunThreads), lds, (hipStream_t)stream, src, dst, g, runs);
        else hipLaunchKernelGGL(morph_runs_kernel<false>, dim3((unsigned)nb), dim3(kRunThreads), lds, (hipStream_t)stream, src, dst, g, runs);
        IPX_LAUNCH_CHECK("morph_runs_kernel");
        return IPX_OK;
    }
    if (op) hipLaunchKernelGGL(morph_kernel<true>, dim3(blocks), dim3(kTileW * kTileH), 0, (hipStream_t)stream, src, dst, g, se);
    else hipLaunchKernelGGL(morph_kernel<false>, dim3(blocks), dim3(kTileW * kTileH), 0, (hipStream_t)stream, src, dst, g, se);
    IPX_LAUNCH_CHECK("morph_kernel");
    return IPX_OK;
}

}  // extern "C"

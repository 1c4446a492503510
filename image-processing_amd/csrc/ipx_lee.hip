// ipx_lee.hip -- Lee speckle filter, server.py:711-732 (reached through mode=denoise_manual, method=lee,
// server.py:224-226).  SURVEY.md section 8(f) row 1.
//
// The reference is a per-pixel Python loop over float32 numpy windows:
//     window = img[y-k:y+k+1, x-k:x+k+1].astype(float32);  mean_w = window.mean();  var_w = window.var()
//     out = mean_w + var_w / (var_w + mean_w**2 / (mean_w + 1e-12)) * (img[y,x] - mean_w)   if var_w > 0 else mean_w
// for interior pixels only (borders keep the input), then clip(0,255).astype(uint8) (truncation).
// To be bit-exact the float32 roundings have to happen in numpy's order:
//   * window.mean(): the sum of <= 961 integers <= 255 is exact in float32 in any order; mean = fl(sum / n).
//   * window.var(): x = fl(v - mean), x = fl(x * x), then numpy's PAIRWISE float32 sum over the n = (2k+1)^2 values in
//     row-major window order (blocks of <= 128 values summed with 8 interleaved accumulators, combined as a tree,
//     remainder added serially; larger n split recursively at n/2 rounded down to a multiple of 8), then fl(sum / n).
// Every float operation below is a separately rounded operation (no FMA contraction, see rn_add etc.).  mean_w**2 on a numpy
// float32 scalar is powf(x, 2.0f); it is taken to be fl(x * x) here (what correctly rounded powf returns).
// One thread per pixel; the block's (8 + 2k) x (32 + 2k) source tile and the window's offset table sit in LDS.
#include "ipx_common.h"
#include <type_traits>

// numpy rounds after every operation: a*b+c must never be fused into one FMA.  In this HIP version the __f*_rn
// intrinsics are plain operators defined in a header, i.e. compiled under the default -ffp-contract=fast, and the
// backend did fuse them (v_fmac_f32 in the variance sum: one result in ~3 million off by one).  So: operators of our own,
// defined under contract(off).
#pragma clang fp contract(off)
namespace {
__device__ __forceinline__ float rn_add(float a, float b) { return a + b; }
__device__ __forceinline__ float rn_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float rn_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float rn_div(float a, float b) { return a / b; }     // IEEE: HIP's default is the correctly rounded one
}  // namespace

namespace {

constexpr int kTileW = 32, kTileH = 8, kMaxK = 15;
constexpr int kPitch = kTileW + 2 * kMaxK + 2;             // 64
constexpr int kMaxN = (2 * kMaxK + 1) * (2 * kMaxK + 1);   // 961

struct LeeGeom {
    int h, w, k;
    int64_t s_row, s_img, d_row, d_img;
    int tiles_x, tiles_y;
};

struct Win {                                               // what a thread needs to read window value i
    const uint8_t* base;                                   // its window's top-left byte in the LDS tile
    const uint16_t* pos;                                   // i -> offset from base
    float mean;
    __device__ __forceinline__ float sq(int i) const
    {
        const float d = rn_sub((float)base[pos[i]], mean);
        return rn_mul(d, d);
    }
};

// numpy's pairwise sum, block case (8 <= n <= 128) and short case (n < 8)
__device__ __forceinline__ float pw_block(const Win& wv, int off, int n)
{
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res = rn_add(res, wv.sq(off + i));
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = wv.sq(off + j);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = rn_add(r[j], wv.sq(off + i + j));
    }
    float res = rn_add(rn_add(rn_add(r[0], r[1]), rn_add(r[2], r[3])),
                          rn_add(rn_add(r[4], r[5]), rn_add(r[6], r[7])));
    for (; i < n; ++i) res = rn_add(res, wv.sq(off + i));
    return res;
}

// recursive case, unrolled to the depth n <= 961 needs (961 -> 480/481 -> 240/241 -> 120/121)
template <int DEPTH>
__device__ __forceinline__ float pw_sum(const Win& wv, int off, int n)
{
    if constexpr (DEPTH == 0) {
        return pw_block(wv, off, n);
    } else {
        if (n <= 128) return pw_block(wv, off, n);
        int n2 = n / 2;
        n2 -= n2 % 8;
        const float a = pw_sum<DEPTH - 1>(wv, off, n2);
        const float b = pw_sum<DEPTH - 1>(wv, off + n2, n - n2);
        return rn_add(a, b);
    }
}

__global__ __launch_bounds__(kTileW * kTileH) void lee_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, LeeGeom g)
{
    __shared__ uint8_t s_tile[(kTileH + 2 * kMaxK) * kPitch];
    __shared__ uint16_t s_pos[kMaxN];
    const int k = g.k, K = 2 * k + 1, n = K * K;
    const unsigned tile = blockIdx.x;
    const int tx0 = (int)(tile % (unsigned)g.tiles_x) * kTileW;
    const int ty0 = (int)(tile / (unsigned)g.tiles_x % (unsigned)g.tiles_y) * kTileH;
    const unsigned img = tile / ((unsigned)g.tiles_x * (unsigned)g.tiles_y);
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    const int tid = (int)threadIdx.x;

    for (int i = tid; i < n; i += kTileW * kTileH) s_pos[i] = (uint16_t)((i / K) * kPitch + i % K);
    const int tw = kTileW + 2 * k, th = kTileH + 2 * k;
    for (int i = tid; i < tw * th; i += kTileW * kTileH) {
        const int ry = i / tw, rx = i - ry * tw;
        int y = ty0 - k + ry, x = tx0 - k + rx;             // out-of-image positions are never used: clamp the address
        y = y < 0 ? 0 : y >= g.h ? g.h - 1 : y;
        x = x < 0 ? 0 : x >= g.w ? g.w - 1 : x;
        s_tile[ry * kPitch + rx] = simg[(int64_t)y * g.s_row + x];
    }
    __syncthreads();

    const int lx = tid % kTileW, ly = tid / kTileW;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= g.w || y >= g.h) return;
    const uint32_t centre = s_tile[(ly + k) * kPitch + lx + k];
    uint32_t outv = centre;
    if (y >= k && y < g.h - k && x >= k && x < g.w - k) {   // interior (server.py:719-720); borders keep the input
        Win wv;
        wv.base = s_tile + ly * kPitch + lx;
        wv.pos = s_pos;
        uint32_t isum = 0;
        for (int i = 0; i < n; ++i) isum += wv.base[s_pos[i]];
        const float fn = (float)n;
        const float mean = rn_div((float)isum, fn);
        wv.mean = mean;
        const float var = rn_div(pw_sum<3>(wv, 0, n), fn);
        float o;
        if (var > 0.f) {
            const float q = rn_div(rn_mul(mean, mean), rn_add(mean, 1e-12f));
            const float k_lee = rn_div(var, rn_add(var, q));
            o = rn_add(mean, rn_mul(k_lee, rn_sub((float)centre, mean)));
        } else {
            o = mean;
        }
        o = o < 0.f ? 0.f : o > 255.f ? 255.f : o;            // np.clip, then astype(uint8) truncates
        outv = (uint32_t)o;
    }
    dst[(int64_t)img * g.d_img + (int64_t)y * g.d_row + x] = (uint8_t)outv;
}


// ---- windows of 3 x 3 and 5 x 5: a thread walks down one image column -------------------------------------------------
// The general kernel above spends its time on LDS index tables and on four IEEE divisions per pixel.  For K = 3 and 5:
//   * a lane owns column x and walks a segment of rows; per row ONE unaligned 4- or 8-byte load gives it bytes
//     x-k .. x+k; the K x K window lives in registers as floats (ring of K rows, loop unrolled by K: static indices);
//   * the window sum S is an integer <= 255 n, so mean = fl(S / n) and q = fl(fl(mean^2) / fl(mean + 1e-12)) are
//     functions of S alone: a (mean, q) table over all S is built per block in LDS with the very same operations
//     (2 divisions per ENTRY instead of per pixel);
//   * the squared deviations are summed in numpy's pairwise order for n < 128 (8 interleaved accumulators over
//     floor(n / 8) rounds, a balanced tree, the remainder serially), all indices static;
//   * two IEEE divisions per pixel remain (sum / n and var / (var + q)).
struct LeeSmallGeom {
    int h, w;
    int64_t s_row, s_img, d_row, d_img;
    int seg_rows, nsegs, nstrips, blocks_per_seg;          // strip = 64 columns (one wave), block = 4 adjacent strips
};

template <int K>
__global__ __launch_bounds__(256) void lee_small_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, LeeSmallGeom g)
{
    constexpr int k = K / 2, n = K * K, NT = 255 * n + 1, LB = K <= 3 ? 4 : 8;   // LB: bytes per row load
    extern __shared__ float2 s_tab[];                      // [S] = (mean, q)
    const float fn = (float)n;
    for (int i = (int)threadIdx.x; i < NT; i += 256) {
        const float m = rn_div((float)i, fn);
        s_tab[i] = make_float2(m, rn_div(rn_mul(m, m), rn_add(m, 1e-12f)));
    }
    __syncthreads();
    const unsigned blk = blockIdx.x;
    const unsigned sub = blk % (unsigned)g.blocks_per_seg, rest = blk / (unsigned)g.blocks_per_seg;
    const unsigned seg = rest % (unsigned)g.nsegs, img = rest / (unsigned)g.nsegs;
    const int x = (int)((sub * 4u + (threadIdx.x >> 6)) * 64u + (threadIdx.x & 63u));
    if (x >= g.w) return;
    const int y0 = (int)seg * g.seg_rows, y1 = y0 + g.seg_rows < g.h ? y0 + g.seg_rows : g.h;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    const bool xin = x >= k && x < g.w - k;                // interior column (server.py:720)
    int a = x - k;                                         // first byte of the row load, kept inside the row (w >= LB)
    a = a > g.w - LB ? g.w - LB : a;
    a = a < 0 ? 0 : a;
    const int sh = xin ? 8 * (x - k - a) : 0;              // window bytes start at bit sh of the load
    const int csh = 8 * (x - a);                           // the centre byte (every column, border ones included)

    typedef typename std::conditional<K <= 3, uint32_t, uint64_t>::type raw_t;
    auto load_row = [&](int t) {                           // walk step t = source row y0 - k + t, clamped into the image
        int yy = y0 - k + t;
        yy = yy < 0 ? 0 : yy > g.h - 1 ? g.h - 1 : yy;
        raw_t v;
        __builtin_memcpy(&v, simg + (int64_t)yy * g.s_row + a, LB);
        return v;
    };
    float f[K][K], rs[K], cen[K];                          // ring of the last K source rows: floats, row sums, centre values
    auto take = [&](int slot, raw_t v) {
        cen[slot] = (float)(uint32_t)((v >> csh) & 255u);
        const raw_t wv = v >> sh;
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < K; ++c) {
            f[slot][c] = (float)(uint32_t)((wv >> (8 * c)) & 255u);
            sum = rn_add(sum, f[slot][c]);                 // integers: exact in any order
        }
        rs[slot] = sum;
    };
    constexpr int U = K;                                   // rows of prefetch
    raw_t nx[U];
#pragma unroll
    for (int j = 0; j < U; ++j) nx[j] = load_row(j);
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
        take(j, nx[j]);
        nx[j] = load_row(U + j);
    }
    const int nrows = y1 - y0;
    for (int t0 = K - 1; t0 - (K - 1) < nrows; t0 += K) {
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int t = t0 + u;
            const int slot = (K - 1 + u) % K;              // = t % K
            take(slot, nx[slot]);
            nx[slot] = load_row(t + U);
            const int yo = t - (K - 1);                    // output row y0 + yo: window rows = ring slots (slot + 1 + r) % K
            if (yo < nrows) {
                const int y = y0 + yo;
                float o = cen[(slot + 1 + k) % K];
                if (xin && y >= k && y < g.h - k) {
                    float S = 0.f;
#pragma unroll
                    for (int r = 0; r < K; ++r) S = rn_add(S, rs[r]);
                    const float2 mq = s_tab[(int)S];
                    const float mean = mq.x;
                    auto sq = [&](int i) {                 // window value i in row-major order
                        const float d = rn_sub(f[(slot + 1 + i / K) % K][i % K], mean);
                        return rn_mul(d, d);
                    };
                    float r8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) r8[j] = sq(j);
#pragma unroll
                    for (int i = 8; i < n - n % 8; i += 8) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) r8[j] = rn_add(r8[j], sq(i + j));
                    }
                    float res = rn_add(rn_add(rn_add(r8[0], r8[1]), rn_add(r8[2], r8[3])),
                                       rn_add(rn_add(r8[4], r8[5]), rn_add(r8[6], r8[7])));
#pragma unroll
                    for (int i = n - n % 8; i < n; ++i) res = rn_add(res, sq(i));
                    const float var = rn_div(res, fn);
                    if (var > 0.f) {
                        const float k_lee = rn_div(var, rn_add(var, mq.y));
                        o = rn_add(mean, rn_mul(k_lee, rn_sub(o, mean)));
                    } else {
                        o = mean;
                    }
                    o = o < 0.f ? 0.f : o > 255.f ? 255.f : o;
                }
                dimg[(int64_t)y * g.d_row + x] = (uint8_t)(uint32_t)o;
            }
        }
    }
}

}  // namespace

extern "C" int ipx_lee_filter_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                                 int64_t src_row_stride, int64_t src_img_stride, int64_t dst_row_stride, int64_t dst_img_stride,
                                 int window_size)
{
    const char* fn = "ipx_lee_filter_u8";
    if (int rc = ipx_check_image_args(fn, src, dst, n, h, w, 1, 1, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride))
        return rc;
    if (window_size < 1 || window_size / 2 > kMaxK)
        return ipx_fail(IPX_ERR_INVALID, "%s: window_size=%d must be in 1..%d", fn, window_size, 2 * kMaxK + 1);
    const int kk = window_size / 2;                        // server.py:717
    if ((kk == 1 || kk == 2) && w >= 64) {                 // 3 x 3 and 5 x 5: the column walker
        LeeSmallGeom sg;
        sg.h = h; sg.w = w;
        sg.s_row = src_row_stride; sg.s_img = src_img_stride; sg.d_row = dst_row_stride; sg.d_img = dst_img_stride;
        sg.nstrips = (w + 63) / 64;
        sg.blocks_per_seg = (sg.nstrips + 3) / 4;
        // segments: enough blocks for a few rounds over the chip, but long enough to amortise the table and the warm-up rows
        int segs = (int)((8192 + (int64_t)sg.blocks_per_seg * n - 1) / ((int64_t)sg.blocks_per_seg * n));
        if (segs > (h + 63) / 64) segs = (h + 63) / 64;
        if (segs < 1) segs = 1;
        sg.seg_rows = (h + segs - 1) / segs;
        sg.nsegs = (h + sg.seg_rows - 1) / sg.seg_rows;
        const uint64_t nb = (uint64_t)n * sg.nsegs * sg.blocks_per_seg;
        if (nb > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: too many blocks", fn);
        const size_t lds = (size_t)(255 * (2 * kk + 1) * (2 * kk + 1) + 1) * sizeof(float2);
        if (kk == 1) hipLaunchKernelGGL(lee_small_kernel<3>, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, src, dst, sg);
        else hipLaunchKernelGGL(lee_small_kernel<5>, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, src, dst, sg);
        IPX_LAUNCH_CHECK("lee_small_kernel");
        return IPX_OK;
    }
    LeeGeom g;
    g.h = h; g.w = w; g.k = kk;
    g.s_row = src_row_stride; g.s_img = src_img_stride; g.d_row = dst_row_stride; g.d_img = dst_img_stride;
    g.tiles_x = (w + kTileW - 1) / kTileW;
    g.tiles_y = (h + kTileH - 1) / kTileH;
    const uint64_t blocks = (uint64_t)n * g.tiles_x * g.tiles_y;
    if (blocks > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: too many tiles", fn);
    hipLaunchKernelGGL(lee_kernel, dim3((unsigned)blocks), dim3(kTileW * kTileH), 0, (hipStream_t)stream, src, dst, g);
    IPX_LAUNCH_CHECK("lee_kernel");
    return IPX_OK;
}

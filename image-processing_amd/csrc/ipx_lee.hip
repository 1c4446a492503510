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

typedef float f2 __attribute__((ext_vector_type(2)));                 // two pixels side by side: v_pk_*_f32
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// fl(a / n) for the constants n = 9, 25, 49 in three operations (Markstein's division by a constant: y = fl(a c) with
// c = fl(1 / n), r = a - n y exactly (one FMA), q = fl(y + r c)).  Correctly rounded for EVERY float a whose quotient neither
// overflows nor is subnormal -- checked exhaustively over all 2^23 significands by tests/test_oracle_golden.py
// (test_markstein_division_by_window_size), and the three steps scale exactly with powers of two.
__device__ __forceinline__ f2 div_by_const(f2 a, float n, float c)
{
    const f2 nn = f2{n, n}, cc = f2{c, c};
    const f2 y = a * cc;
    return fma2(fma2(-nn, y, a), cc, y);
}

// fl(a / b) for the two divisions of the Lee filter: the gain var / (var + q) (2^-60 < a <= b, 2^-8 < b < 2^20) and, in the forms
// without a table, q = fl(mean^2) / fl(mean + 1e-12) for window sums S > 0 (2^-14 < a < 2^16, 2^-7 < b < 2^8, quotient < 2^8).
// It is the operation sequence of the hardware's own IEEE division (v_div_scale, v_rcp, 4 FMAs + 1 multiply for the refinement,
// v_div_fmas, v_div_fixup) with the parts that only act on extreme exponents, zeros, infinities and NaNs left out: in both ranges
// the operands' and the quotient's exponents stay far from where v_div_scale scales and v_div_fixup intervenes, so the result is
// bit-identical to a / b.  tests/test_oracle_golden.py runs the sequence (fmaf on the CPU, the reciprocal seed also one ulp off
// either way) over EVERY window sum of 49, 81 and 121 values and a sweep of gains; every Lee parity test goes through it on the GPU.
__device__ __forceinline__ f2 div_safe_range(f2 a, f2 b)
{
    f2 r = f2{__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
    r = fma2(fma2(-b, r, f2{1.0f, 1.0f}), r, r);
    f2 q = a * r;
    q = fma2(fma2(-b, q, a), r, q);
    return fma2(fma2(-b, q, a), r, q);
}
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


// ---- windows of 11 x 11 and up: two adjacent pixels per thread, packed float32 ---------------------------------------------
// The kernel above pays, per window value and pixel, two LDS reads (offset table, byte), a conversion and three float
// operations, twice over (sum, then squared deviations): ~12 instructions.  Here a thread owns pixels x, x + 1, whose
// windows are one column apart: the tile sits in LDS as FLOATS, so window value i of both pixels is one ds_read2_b32 of two
// neighbouring floats -- a 64-bit register pair, the operand of v_pk_add_f32 / v_pk_mul_f32 -- and the three operations per
// value (x - mean, squared, added to its accumulator; each separately rounded, numpy's pairwise order exactly as above)
// are three packed instructions for the two pixels.  The window offset of value i is kept in scalar registers (a cursor
// that walks the window in row-major order, which is the order the pairwise sum consumes the values in); the kernel is
// compiled per window size, the loops unroll completely and the cursor becomes immediate offsets (neighbouring values'
// reads merge: one float is read once per thread).  The integer window sums (exact in float32 in any order) come from
// horizontal K-sums of the tile rows, built once per block in LDS: K reads per pixel pair instead of K * K.
constexpr int kWideW = 64, kWideH = 8, kWidePitch = kWideW + 2 * kMaxK + 2;          // 96 floats per tile row

template <int K>
struct WideWin {
    const float* base;                                     // the left pixel's window, top-left float in the LDS tile
    int cur, col;                                          // cursor: float offset of the next window value, its column
    f2 mean;                                               // (compile-time constants once the loops below are unrolled)
    __device__ __forceinline__ f2 next_sq()
    {
        const f2 v = f2{base[cur], base[cur + 1]};
        ++col;
        const bool wrap = col == K;
        cur += wrap ? kWidePitch - K + 1 : 1;
        col = wrap ? 0 : col;
        const f2 d = v - mean;
        return d * d;
    }
};

// numpy's pairwise sum over the next N values of the cursor (the same three cases as pw_block / pw_sum above), fully unrolled
template <int K, int N>
__device__ __forceinline__ f2 pw_block2(WideWin<K>& wv)
{
    if constexpr (N < 8) {
        f2 res = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < N; ++i) res = res + wv.next_sq();
        return res;
    } else {
        f2 r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = wv.next_sq();
#pragma unroll
        for (int i = 8; i < N - (N % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = r[j] + wv.next_sq();
        }
        f2 res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int i = N - (N % 8); i < N; ++i) res = res + wv.next_sq();
        return res;
    }
}

template <int K, int N>
__device__ __forceinline__ f2 pw_sum2(WideWin<K>& wv)
{
    if constexpr (N <= 128) {
        return pw_block2<K, N>(wv);
    } else {
        constexpr int N2 = N / 2 - (N / 2) % 8;
        const f2 a = pw_sum2<K, N2>(wv);
        const f2 b = pw_sum2<K, N - N2>(wv);
        return a + b;
    }
}

template <int K>
__global__ __launch_bounds__(kWideW / 2 * kWideH) void lee_wide_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, LeeGeom g)
{
    __shared__ float s_tile[(kWideH + 2 * kMaxK) * kWidePitch];
    __shared__ float s_hsum[(kWideH + 2 * kMaxK) * kWideW];  // s_hsum[r][c] = tile[r][c] + ... + tile[r][c + K - 1]
    constexpr int kThreads = kWideW / 2 * kWideH;
    constexpr int k = K / 2, n = K * K;
    const unsigned tile = blockIdx.x;
    const int tx0 = (int)(tile % (unsigned)g.tiles_x) * kWideW;
    const int ty0 = (int)(tile / (unsigned)g.tiles_x % (unsigned)g.tiles_y) * kWideH;
    const unsigned img = tile / ((unsigned)g.tiles_x * (unsigned)g.tiles_y);
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    const int tid = (int)threadIdx.x;

    const int tw = kWideW + 2 * k, th = kWideH + 2 * k;
    for (int i = tid; i < tw * th; i += kThreads) {
        const int ry = i / tw, rx = i - ry * tw;
        int y = ty0 - k + ry, x = tx0 - k + rx;             // out-of-image positions are never used: clamp the address
        y = y < 0 ? 0 : y >= g.h ? g.h - 1 : y;
        x = x < 0 ? 0 : x >= g.w ? g.w - 1 : x;
        s_tile[ry * kWidePitch + rx] = (float)simg[(int64_t)y * g.s_row + x];
    }
    if (tid < th) s_tile[tid * kWidePitch + tw] = 0.f;      // the right pixel of the last pair reads one float past the row
    __syncthreads();
    // horizontal K-sums of every tile row (integers < 2^24: exact in float32 in any order), a running sum per 8 columns
    for (int t = tid; t < th * (kWideW / 8); t += kThreads) {
        const int r = t / (kWideW / 8), c0 = (t % (kWideW / 8)) * 8;
        const float* row = s_tile + r * kWidePitch + c0;
        float sum = 0.f;
        for (int j = 0; j < K; ++j) sum += row[j];
        float* hp = s_hsum + r * kWideW + c0;
        hp[0] = sum;
#pragma unroll
        for (int c = 1; c < 8; ++c) {
            sum += row[c - 1 + K] - row[c - 1];
            hp[c] = sum;
        }
    }
    __syncthreads();

    const int lx = 2 * (tid % (kWideW / 2)), ly = tid / (kWideW / 2);
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= g.w || y >= g.h) return;
    const float* base = s_tile + ly * kWidePitch + lx;
    const f2 centre = f2{base[k * kWidePitch + k], base[k * kWidePitch + k + 1]};
    f2 o = centre;
    const bool rows_in = y >= k && y < g.h - k;
    const bool in0 = rows_in && x >= k && x < g.w - k, in1 = rows_in && x + 1 >= k && x + 1 < g.w - k;
    if (in0 || in1) {                                       // interior (server.py:719-720); borders keep the input
        f2 isum = {0.f, 0.f};                               // the window sums: K row sums each
        const float* hs = s_hsum + ly * kWideW + lx;
#pragma unroll
        for (int r = 0; r < K; ++r) isum = isum + f2{hs[r * kWideW], hs[r * kWideW + 1]};
        const float fn = (float)n;
        WideWin<K> wv;
        wv.base = base; wv.cur = 0; wv.col = 0;
        wv.mean = f2{rn_div(isum.x, fn), rn_div(isum.y, fn)};
        const f2 ssq = pw_sum2<K, n>(wv);
        const float mean2[2] = {wv.mean.x, wv.mean.y}, ss[2] = {ssq.x, ssq.y}, cen[2] = {centre.x, centre.y};
        float res[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float mean = mean2[p], var = rn_div(ss[p], fn);
            float v;
            if (var > 0.f) {
                const float q = rn_div(rn_mul(mean, mean), rn_add(mean, 1e-12f));
                const float k_lee = rn_div(var, rn_add(var, q));
                v = rn_add(mean, rn_mul(k_lee, rn_sub(cen[p], mean)));
            } else {
                v = mean;
            }
            res[p] = v < 0.f ? 0.f : v > 255.f ? 255.f : v;     // np.clip, then astype(uint8) truncates
        }
        if (in0) o.x = res[0];
        if (in1) o.y = res[1];
    }
    uint8_t* dp = dst + (int64_t)img * g.d_img + (int64_t)y * g.d_row + x;
    dp[0] = (uint8_t)(uint32_t)o.x;
    if (x + 1 < g.w) dp[1] = (uint8_t)(uint32_t)o.y;
}


// ---- windows of 3 x 3, 5 x 5 and 7 x 7: a thread walks down two adjacent image columns ----------------------------------------
// The general kernel above spends its time on LDS index tables and on four IEEE divisions per pixel.  For K = 3, 5 and 7:
//   * a lane owns columns x, x + 1 and walks a segment of rows; per row ONE unaligned 4- or 8-byte load gives it bytes
//     x-k .. x+1+k; the two K x K windows live in registers as PAIRS of floats (window value i of the left and of the
//     right pixel side by side), ring of K rows, loop unrolled by K: static indices.  All per-pixel arithmetic is then
//     two-wide (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32), every operation separately rounded as before;
//   * the window sum S is an integer <= 255 n, so mean = fl(S / n) and q = fl(fl(mean^2) / fl(mean + 1e-12)) are
//     functions of S alone: a (mean, q) table over all S is built per block in LDS with the very same operations
//     (2 divisions per ENTRY instead of per pixel); K = 7 computes both per pixel instead (its table would not fit);
//   * the squared deviations are summed in numpy's pairwise order for n < 128 (8 interleaved accumulators over
//     floor(n / 8) rounds, a balanced tree, the remainder serially), all indices static;
//   * the two divisions per pixel that remain are sum / n (three operations, div_by_const) and var / (var + q) (the
//     hardware division's own refinement steps without its range handling, div_safe_range).
struct LeeSmallGeom {
    int h, w;
    int64_t s_row, s_img, d_row, d_img;
    int seg_rows, nsegs, nstrips, blocks_per_seg;          // strip = 128 columns (one wave), block = 4 adjacent strips
};

template <int K>
__global__ __launch_bounds__(256) void lee_small_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, LeeSmallGeom g)
{
    constexpr int k = K / 2, n = K * K, NT = 255 * n + 1, LB = K <= 3 ? 4 : K <= 7 ? 8 : 12;   // LB: bytes per row load (>= K + 1)
    constexpr bool TABLE = K <= 5;                         // K = 7: the table would take 100 KB of LDS; mean and q are computed
    extern __shared__ float s_tab[];                       // mean[S] = s_tab[S], q[S] = s_tab[NT + S]
    const float fn = (float)n, cn = rn_div(1.0f, fn);
    if (TABLE) {
        for (int i = (int)threadIdx.x; i < NT; i += 256) {
            const float m = rn_div((float)i, fn);
            s_tab[i] = m;
            // S = 0 (an all-zero window: var = 0, mean = 0) would give q = 0 and 0 / 0 below; any positive q there gives the
            // reference's result (k_lee = 0, out = mean)
            s_tab[NT + i] = i ? rn_div(rn_mul(m, m), rn_add(m, 1e-12f)) : 1.0f;
        }
        __syncthreads();
    }
    const unsigned blk = blockIdx.x;
    const unsigned sub = blk % (unsigned)g.blocks_per_seg, rest = blk / (unsigned)g.blocks_per_seg;
    const unsigned seg = rest % (unsigned)g.nsegs, img = rest / (unsigned)g.nsegs;
    const int x = 2 * (int)((sub * 4u + (threadIdx.x >> 6)) * 64u + (threadIdx.x & 63u));
    if (x >= g.w) return;
    const int y0 = (int)seg * g.seg_rows, y1 = y0 + g.seg_rows < g.h ? y0 + g.seg_rows : g.h;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    const bool in0 = x >= k && x < g.w - k, in1 = x + 1 >= k && x + 1 < g.w - k;   // interior columns (server.py:720)
    const bool both = in0 && in1;
    int a = x - k;                                         // first byte of the row load, kept inside the row (w >= LB)
    a = a > g.w - LB ? g.w - LB : a;
    a = a < 0 ? 0 : a;
    // window byte j (column x - k + j) is byte j + d of the load, d = x - k - a (negative in the row's first lane): byte
    // selectors for v_perm_b32, 0x0c = constant zero for bytes the load does not hold (never used by an interior pixel)
    const int d = x - k - a;
    // (LB = 12: three dwords are more than one v_perm_b32 can select from; there the window bytes come from the load by a
    // per-lane byte shift, v_alignbyte_b32 over [0, 0, w0, w1, w2, 0, 0], the dword part of the shift chosen by compares)
    const uint32_t shb = (uint32_t)(d + 8) & 3u, shd = (uint32_t)(d + 8) >> 2;     // 3 <= d + 8 <= LB - 1 - k + 8 <= 15
    uint32_t sel[LB / 4];
#pragma unroll
    for (int q = 0; q < LB / 4; ++q) {
        uint32_t sq_ = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sb = 4 * q + j + d;
            sq_ |= (uint32_t)(sb >= 0 && sb < LB ? sb : 0x0c) << (8 * j);
        }
        sel[q] = sq_;
    }
    const uint32_t aoff = (uint32_t)a, xoff = (uint32_t)x;
    const bool even_w = (g.w & 1) == 0;                    // every lane then owns two columns

    struct Raw { uint32_t w[LB / 4]; };
    auto load_row = [&](int t) {                           // walk step t = source row y0 - k + t, clamped into the image
        int yy = y0 - k + t;
        yy = yy < 0 ? 0 : yy > g.h - 1 ? g.h - 1 : yy;
        const uint8_t* rowp = simg + (int64_t)yy * g.s_row;   // wave-uniform: scalar base + 32-bit lane offset
        Raw v;
        __builtin_memcpy(&v, rowp + aoff, LB);
        return v;
    };
    f2 P[K][K], rs[K];                                     // ring of the last K source rows: value pairs, row sums
    auto take = [&](int slot, const Raw& v) {
        float fl[K + 1];
#pragma unroll
        for (int q = 0; q < LB / 4; ++q) {
            uint32_t wv;
            if (LB <= 8) {
                wv = __builtin_amdgcn_perm(LB == 8 ? v.w[LB / 4 - 1] : 0u, v.w[0], sel[q]);
            } else {
                const uint32_t e[9] = {0u, 0u, v.w[0], v.w[LB >= 8 ? 1 : 0], v.w[LB >= 12 ? 2 : 0], 0u, 0u, 0u, 0u};
                const uint32_t c0 = __builtin_amdgcn_alignbyte(e[q + 1], e[q], shb), c1 = __builtin_amdgcn_alignbyte(e[q + 2], e[q + 1], shb),
                               c2 = __builtin_amdgcn_alignbyte(e[q + 3], e[q + 2], shb), c3 = __builtin_amdgcn_alignbyte(e[q + 4], e[q + 3], shb);
                wv = shd == 0 ? c0 : shd == 1 ? c1 : shd == 2 ? c2 : c3;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * q + j <= K) fl[4 * q + j] = (float)((wv >> (8 * j)) & 255u);
        }
        f2 sum = f2{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < K; ++c) {
            P[slot][c] = f2{fl[c], fl[c + 1]};             // P[.][k] is the pair of centre values, border columns included
            sum = sum + P[slot][c];                        // integers: exact in any order
        }
        rs[slot] = sum;
    };
    Raw nx[K];
#pragma unroll
    for (int j = 0; j < K; ++j) nx[j] = load_row(j);
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
        take(j, nx[j]);
        nx[j] = load_row(K + j);
    }
    const int nrows = y1 - y0;
    for (int t0 = K - 1; t0 - (K - 1) < nrows; t0 += K) {
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int t = t0 + u;
            const int slot = (K - 1 + u) % K;              // = t % K
            take(slot, nx[slot]);
            nx[slot] = load_row(t + K);
            const int yo = t - (K - 1);                    // output row y0 + yo: window rows = ring slots (slot + 1 + r) % K
            if (yo < nrows) {
                const int y = y0 + yo;
                const f2 c2 = P[(slot + 1 + k) % K][k];
                f2 o = c2;
                if (y >= k && y < g.h - k) {               // interior row (wave-uniform)
                    f2 S = rs[0];
#pragma unroll
                    for (int r = 1; r < K; ++r) S = S + rs[r];
                    f2 mean, q;
                    if (TABLE) {
                        const int i0 = (int)S.x, i1 = (int)S.y;
                        mean = f2{s_tab[i0], s_tab[i1]};
                        q = f2{s_tab[NT + i0], s_tab[NT + i1]};
                    } else {                               // the same values by the same roundings: S / n in three operations,
                        mean = div_by_const(S, fn, cn);    // fl(mean^2) / fl(mean + 1e-12) with operands >= 1 / n (S = 0: q := 1)
                        const f2 e12 = f2{1e-12f, 1e-12f};
                        const f2 qq = div_safe_range(mean * mean, mean + e12);
                        q = f2{S.x > 0.f ? qq.x : 1.0f, S.y > 0.f ? qq.y : 1.0f};
                    }
                    auto sq = [&](int i) {                 // window value i in row-major order, both pixels
                        const f2 dv = P[(slot + 1 + i / K) % K][i % K] - mean;
                        return dv * dv;
                    };
                    f2 r8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) r8[j] = sq(j);
#pragma unroll
                    for (int i = 8; i < n - n % 8; i += 8) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) r8[j] = r8[j] + sq(i + j);
                    }
                    f2 res = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
#pragma unroll
                    for (int i = n - n % 8; i < n; ++i) res = res + sq(i);
                    const f2 var = div_by_const(res, fn, cn);
                    // var == 0 needs no branch: k_lee = 0 / q = 0 and mean + 0 * (c - mean) = mean, the reference's other arm.
                    // np.clip needs no code either: 0 <= k_lee <= 1 and every rounding is monotonic, so the result lies
                    // between mean and the centre value (within a rounding of it), i.e. in [0, 256) and truncates the same
                    const f2 kl = div_safe_range(var, var + q);
                    const f2 lee = mean + kl * (c2 - mean);
                    if (both) o = lee;
                    else o = f2{in0 ? lee.x : c2.x, in1 ? lee.y : c2.y};
                }
                uint8_t* dp = dimg + (int64_t)y * g.d_row;  // wave-uniform
                const uint32_t b0 = (uint32_t)o.x, b1 = (uint32_t)o.y;
                if (even_w || x + 1 < g.w) {
                    const uint16_t two = (uint16_t)(b0 | (b1 << 8));
                    __builtin_memcpy(dp + xoff, &two, 2);
                } else {
                    dp[xoff] = (uint8_t)b0;
                }
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
    if (window_size / 2 > 0)
        if (int rc = ipx_check_disjoint(fn, "source", src, w, src_row_stride, src_img_stride, "destination", dst, w, dst_row_stride, dst_img_stride, n, h))
            return rc;
    const int kk = window_size / 2;                        // server.py:717
    if (kk >= 1 && kk <= 4 && w >= 64) {                   // 3 x 3 ... 9 x 9: the column walker (11 x 11: lee_wide_kernel is faster,
                                                           // 2.14 against 2.90 ms per 24 x 4K; 9 x 9: the walker, 1.43 against 1.85)
        LeeSmallGeom sg;
        sg.h = h; sg.w = w;
        sg.s_row = src_row_stride; sg.s_img = src_img_stride; sg.d_row = dst_row_stride; sg.d_img = dst_img_stride;
        sg.nstrips = (w + 127) / 128;
        sg.blocks_per_seg = (sg.nstrips + 3) / 4;
        // segments: enough blocks for a few rounds over the chip, but long enough to amortise the table and the warm-up rows
        int segs = (int)((8192 + (int64_t)sg.blocks_per_seg * n - 1) / ((int64_t)sg.blocks_per_seg * n));
        if (segs > (h + 63) / 64) segs = (h + 63) / 64;
        if (segs < 1) segs = 1;
        sg.seg_rows = (h + segs - 1) / segs;
        {   // small batches (one 4K frame: 270 blocks for 768 to 2048 block slots): the segment length that minimises rounds of
            // resident blocks x rows a block works through (its rows, the K - 1 rows above them, ~4 rows' worth of table)
            const int K = 2 * kk + 1, extra = K - 1 + (kk <= 2 ? 4 : 0);
            const int64_t slots = 256 * (kk == 1 ? 8 : kk <= 3 ? 3 : 1), per_seg = (int64_t)n * sg.blocks_per_seg;
            if (per_seg * ((h + sg.seg_rows - 1) / sg.seg_rows) < 2 * slots) {
                int64_t best_cost = INT64_MAX;
                int best = sg.seg_rows;
                for (int seg = K; seg <= sg.seg_rows; seg += K) {
                    const int64_t blocks = per_seg * ((h + seg - 1) / seg);
                    const int64_t cost = ((blocks + slots - 1) / slots) * (seg + extra);
                    if (cost < best_cost) { best_cost = cost; best = seg; }
                }
                sg.seg_rows = best;
            }
        }
        sg.nsegs = (h + sg.seg_rows - 1) / sg.seg_rows;
        const uint64_t nb = (uint64_t)n * sg.nsegs * sg.blocks_per_seg;
        if (nb > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: too many blocks", fn);
        const size_t lds = kk <= 2 ? (size_t)(255 * (2 * kk + 1) * (2 * kk + 1) + 1) * 2 * sizeof(float) : 0;
        if (kk == 1) hipLaunchKernelGGL(lee_small_kernel<3>, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, src, dst, sg);
        else if (kk == 2) hipLaunchKernelGGL(lee_small_kernel<5>, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, src, dst, sg);
        else if (kk == 3) hipLaunchKernelGGL(lee_small_kernel<7>, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, src, dst, sg);
        else hipLaunchKernelGGL(lee_small_kernel<9>, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, src, dst, sg);
        IPX_LAUNCH_CHECK("lee_small_kernel");
        return IPX_OK;
    }
    LeeGeom g;
    g.h = h; g.w = w; g.k = kk;
    g.s_row = src_row_stride; g.s_img = src_img_stride; g.d_row = dst_row_stride; g.d_img = dst_img_stride;
    if (kk >= 5) {                                         // 11 x 11 and up: two pixels per thread, packed float32
        g.tiles_x = (w + kWideW - 1) / kWideW;
        g.tiles_y = (h + kWideH - 1) / kWideH;
        const uint64_t blocks = (uint64_t)n * g.tiles_x * g.tiles_y;
        if (blocks > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: too many tiles", fn);
        const dim3 grid((unsigned)blocks), block(kWideW / 2 * kWideH);
        hipStream_t st = (hipStream_t)stream;
        switch (kk) {
        case 5: hipLaunchKernelGGL(lee_wide_kernel<11>, grid, block, 0, st, src, dst, g); break;
        case 6: hipLaunchKernelGGL(lee_wide_kernel<13>, grid, block, 0, st, src, dst, g); break;
        case 7: hipLaunchKernelGGL(lee_wide_kernel<15>, grid, block, 0, st, src, dst, g); break;
        case 8: hipLaunchKernelGGL(lee_wide_kernel<17>, grid, block, 0, st, src, dst, g); break;
        case 9: hipLaunchKernelGGL(lee_wide_kernel<19>, grid, block, 0, st, src, dst, g); break;
        case 10: hipLaunchKernelGGL(lee_wide_kernel<21>, grid, block, 0, st, src, dst, g); break;
        case 11: hipLaunchKernelGGL(lee_wide_kernel<23>, grid, block, 0, st, src, dst, g); break;
        case 12: hipLaunchKernelGGL(lee_wide_kernel<25>, grid, block, 0, st, src, dst, g); break;
        case 13: hipLaunchKernelGGL(lee_wide_kernel<27>, grid, block, 0, st, src, dst, g); break;
        case 14: hipLaunchKernelGGL(lee_wide_kernel<29>, grid, block, 0, st, src, dst, g); break;
        default: hipLaunchKernelGGL(lee_wide_kernel<31>, grid, block, 0, st, src, dst, g); break;
        }
        IPX_LAUNCH_CHECK("lee_wide_kernel");
        return IPX_OK;
    }
    g.tiles_x = (w + kTileW - 1) / kTileW;
    g.tiles_y = (h + kTileH - 1) / kTileH;
    const uint64_t blocks = (uint64_t)n * g.tiles_x * g.tiles_y;
    if (blocks > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: too many tiles", fn);
    hipLaunchKernelGGL(lee_kernel, dim3((unsigned)blocks), dim3(kTileW * kTileH), 0, (hipStream_t)stream, src, dst, g);
    IPX_LAUNCH_CHECK("lee_kernel");
    return IPX_OK;
}

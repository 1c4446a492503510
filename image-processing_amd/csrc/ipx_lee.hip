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
    LeeGeom g;
    g.h = h; g.w = w; g.k = window_size / 2;               // server.py:717
    g.s_row = src_row_stride; g.s_img = src_img_stride; g.d_row = dst_row_stride; g.d_img = dst_img_stride;
    g.tiles_x = (w + kTileW - 1) / kTileW;
    g.tiles_y = (h + kTileH - 1) / kTileH;
    const uint64_t blocks = (uint64_t)n * g.tiles_x * g.tiles_y;
    if (blocks > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: too many tiles", fn);
    hipLaunchKernelGGL(lee_kernel, dim3((unsigned)blocks), dim3(kTileW * kTileH), 0, (hipStream_t)stream, src, dst, g);
    IPX_LAUNCH_CHECK("lee_kernel");
    return IPX_OK;
}

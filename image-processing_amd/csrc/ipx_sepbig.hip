// ipx_sepbig.hip -- separable fixed-point filter for 9 <= k <= 31 taps (cv2.GaussianBlur with a large kernel,
// server.py:428; BASELINE.json configs[3]: 15x15 and 31x31 on 8K).
//
// At 2k multiply-adds per byte a 31-tap filter is VALU work as much as memory work, and a K-row ring of Q8.8 row
// results (K x 2 KiB per wave) no longer fits registers or LDS at a useful occupancy.  So the two passes are two
// kernels around a Q8.8 scratch image:
//   sep_h_kernel  row pass, a FLAT stream (no vertical dependency): every lane turns 16 bytes of one row (plus its
//                 halo window) into 16 Q8.8 results with packed 16-bit multiply-adds (v_pk_mad_u16) and stores them
//                 in "field order" (the packed pairs as they are, 32 B per lane) -- the column pass works per byte
//                 column, so it never needs them in natural order.  Rows -r .. h+r-1 are produced (REFLECT_101 or
//                 the caller's halo rows), so the column pass has no border logic at all.
//   sep_v_kernel  column pass: a wave walks a strip (64 lanes x 8 columns) in blocks of RB = 16 output rows; each
//                 scratch row is loaded once per block and feeds up to 16 accumulator rows (v_mad_u32_u16 on the
//                 lo/hi half, taps as scalars), so a row is re-read (k + 15) / 16 times.
// Taps are padded with zeros to the kernel's class size (k <= 15 or k <= 31): zero taps cost work, not exactness.
#include "ipx_common.h"
#include "ipx_fields.h"

namespace {

constexpr int kThreads = 256;
constexpr int RB = 16;                                  // output rows per column-pass block

struct HGeom {
    int h, w, L, r;                                     // image rows, pixels, row bytes; real radius
    int rows_tmp;                                       // h + 2r scratch rows per image
    int pitch;                                          // scratch row pitch in u16 elements (multiple of 16)
    int64_t s_row, s_img;
    int off;                                            // main region starts at byte `off` (multiple of 16, >= halo)
    int nlanes;                                         // 16-byte lanes of the main region
    int nb;                                             // border bytes per row
    unsigned main_items, main_blocks, border_total;     // lane-rows, blocks of them, border bytes in all
    const uint8_t* top;
    const uint8_t* bot;
};

template <int RMAX> struct TapsPad { uint32_t k[2 * RMAX + 1]; };   // centred, zero padded

__device__ __forceinline__ const uint8_t* src_row(const HGeom& g, const uint8_t* simg, int y)
{
    return halo_row(simg, g.top, g.bot, y, g.h, g.r, g.s_row);
}

// u16 index, inside a scratch row, of the row-pass result for byte b (field order within each 16-byte group)
__device__ __forceinline__ int field_index(int b)
{
    const int j = b & 15, k = j >> 2, m = j & 3;
    return (b & ~15) + 2 * (2 * k + (m & 1)) + (m >> 1);
}

template <int C, int RMAX>
__global__ __launch_bounds__(kThreads) void sep_h_kernel(const uint8_t* __restrict__ src, uint16_t* __restrict__ tmp,
                                                         HGeom g, TapsPad<RMAX> t)
{
    constexpr int K = 2 * RMAX + 1;
    if (blockIdx.x >= g.main_blocks) {                  // ---- border bytes: one per thread, pixel-space reflection
        const unsigned id = (blockIdx.x - g.main_blocks) * kThreads + threadIdx.x;
        if (id >= g.border_total) return;
        const unsigned j = id % (unsigned)g.nb, rest = id / (unsigned)g.nb;
        const int trow = (int)(rest % (unsigned)g.rows_tmp);
        const unsigned img = rest / (unsigned)g.rows_tmp;
        const int b = g.nlanes > 0 ? ((int)j < g.off ? (int)j : 16 * g.nlanes + (int)j) : (int)j;
        const uint8_t* rp = src_row(g, src + (int64_t)img * g.s_img, trow - g.r);
        const int px = b / C, ch = b - px * C;
        uint32_t acc = 0;
        for (int q = RMAX - g.r; q <= RMAX + g.r; ++q)   // only the real taps: the padding is zero
            acc += t.k[q] * rp[reflect101(px + q - RMAX, g.w) * C + ch];
        tmp[((int64_t)img * g.rows_tmp + trow) * g.pitch + field_index(b)] = (uint16_t)acc;
        return;
    }
    const unsigned item = blockIdx.x * kThreads + threadIdx.x;
    if (item >= g.main_items) return;
    const unsigned lane = item % (unsigned)g.nlanes, rest = item / (unsigned)g.nlanes;
    const int trow = (int)(rest % (unsigned)g.rows_tmp);
    const unsigned img = rest / (unsigned)g.rows_tmp;
    const int b0 = g.off + 16 * (int)lane;
    Win<C, RMAX> W;
    load_win<C, RMAX>(W, src_row(g, src + (int64_t)img * g.s_img, trow - g.r), b0, g.L);
    us2 h[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int p = slot_p(s);
        us2 acc = fld<C, RMAX>(W, p - C * RMAX) * (unsigned short)t.k[0];
#pragma unroll
        for (int q = 1; q < K; ++q) acc += fld<C, RMAX>(W, p + C * (q - RMAX)) * (unsigned short)t.k[q];
        h[s] = acc;                                     // Q8.8 <= 255 * 256
    }
    u32x4 lo, hi;
#pragma unroll
    for (int i = 0; i < 4; ++i) { lo[i] = __builtin_bit_cast(uint32_t, h[i]); hi[i] = __builtin_bit_cast(uint32_t, h[4 + i]); }
    uint8_t* out = reinterpret_cast<uint8_t*>(tmp + ((int64_t)img * g.rows_tmp + trow) * g.pitch + b0);
    st16(out, lo);
    st16(out + 16, hi);
}

struct VGeom {
    int h, L, pitch, rows_tmp, K;
    int64_t d_row, d_img;
    int nhalf, nstrips, seg_rows, nsegs;                  // nhalf: 8-column half groups per row
    unsigned nitems;
};

struct TapsV { uint32_t k[96]; };                         // (RB-1) zeros, K taps, zeros up to the loop's reach

// Column pass.  A lane owns 8 byte columns (16 B of a scratch row: 4 packed field pairs) and RB = 16 output rows at
// a time: every scratch row of the block is loaded once and feeds the up-to-16 accumulator rows it belongs to, so a
// row is re-read (K + 15) / 16 times instead of K times.  Taps are scalars (SGPRs), the tap of (row i0+u, output r)
// is c[u - r + RB - 1]; the host pads the tap table with zeros so no index is ever tested.
__global__ __launch_bounds__(kThreads) void sep_v_kernel(const uint16_t* __restrict__ tmp, uint8_t* __restrict__ dst,
                                                         VGeom g, TapsV ky)
{
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned item = xcd_remap(blockIdx.x, gridDim.x) * (kThreads / 64) + wave;
    if (item >= g.nitems) return;
    const unsigned strip = item % (unsigned)g.nstrips, rest = item / (unsigned)g.nstrips;
    const unsigned seg = rest % (unsigned)g.nsegs, img = rest / (unsigned)g.nsegs;
    const int half = (int)strip * 64 + (int)(threadIdx.x & 63u);
    if (half >= g.nhalf) return;
    const int y0 = (int)seg * g.seg_rows;
    const int y1 = y0 + g.seg_rows < g.h ? y0 + g.seg_rows : g.h;
    const uint8_t* timg = reinterpret_cast<const uint8_t*>(tmp + (int64_t)img * g.rows_tmp * g.pitch + 8 * half);
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    const int64_t tpitch = (int64_t)g.pitch * 2;
    const int b0 = 8 * half;
    const int last = g.rows_tmp - 1;

    for (int yb = y0; yb < y1; yb += RB) {
        uint32_t al[RB][4], ah[RB][4];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int s = 0; s < 4; ++s) { al[r][s] = 32768u; ah[r][s] = 32768u; }
        for (int i0 = 0; i0 < g.K + RB - 1; i0 += RB) {
            uint32_t c[2 * RB - 1];                       // taps i0 - (RB-1) .. i0 + (RB-1)
#pragma unroll
            for (int d = 0; d < 2 * RB - 1; ++d) c[d] = __builtin_amdgcn_readfirstlane(ky.k[i0 + d]);
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int yy = yb + i0 + u < last ? yb + i0 + u : last;   // rows nobody needs: clamped, taps are 0
                const u32x4 row = ld16(timg + (int64_t)yy * tpitch);
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const uint32_t cf = c[u - r + RB - 1];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        al[r][s] = mad_lo16(row[s], cf, al[r][s]);
                        ah[r][s] = mad_hi16(row[s], cf, ah[r][s]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if (yb + r >= y1) break;
            u32x2 o;
#pragma unroll
            for (int k = 0; k < 2; ++k) o[k] = pack_acc(al[r][2 * k], al[r][2 * k + 1], ah[r][2 * k], ah[r][2 * k + 1]);
            uint8_t* dp = dimg + (int64_t)(yb + r) * g.d_row;
            if (b0 + 8 <= g.L) {
                __builtin_nontemporal_store(o, (u32x2_u*)(dp + b0));
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (b0 + q < g.L) dp[b0 + q] = (uint8_t)(o[q >> 2] >> (8 * (q & 3)));
            }
        }
    }
}

template <int C, int RMAX>
void launch_h(hipStream_t st, const uint8_t* src, uint16_t* tmp, const HGeom& g, const uint16_t* kx, int k)
{
    TapsPad<RMAX> t{};
    for (int i = 0; i < k; ++i) t.k[RMAX - k / 2 + i] = kx[i];
    const unsigned grid = g.main_blocks + (g.border_total + kThreads - 1) / kThreads;
    hipLaunchKernelGGL((sep_h_kernel<C, RMAX>), dim3(grid), dim3(kThreads), 0, st, src, tmp, g, t);
}

template <int RMAX>
void launch_h_c(hipStream_t st, const uint8_t* src, uint16_t* tmp, const HGeom& g, int c, const uint16_t* kx, int k)
{
    switch (c) {
        case 1: launch_h<1, RMAX>(st, src, tmp, g, kx, k); break;
        case 2: launch_h<2, RMAX>(st, src, tmp, g, kx, k); break;
        case 3: launch_h<3, RMAX>(st, src, tmp, g, kx, k); break;
        default: launch_h<4, RMAX>(st, src, tmp, g, kx, k); break;
    }
}

}  // namespace

// Called by ipx_sep_filter_q8_u8 for 9 <= k <= 31 (arguments already validated there).
int ipx_sep_mid(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                const uint16_t* kx, const uint16_t* ky, int k, const uint8_t* top, const uint8_t* bot)
{
    const char* fn = "ipx_sep_filter_q8_u8";
    const int r = k / 2, rmax = k <= 15 ? 7 : 15;
    HGeom g;
    g.h = h; g.w = w; g.L = w * c; g.r = r; g.rows_tmp = h + 2 * r;
    g.pitch = (g.L + 15) & ~15;
    g.s_row = s_row; g.s_img = s_img; g.top = top; g.bot = bot;
    const int hd4 = 4 * ((rmax * c + 3) / 4);
    g.off = (hd4 + 15) & ~15;
    g.nlanes = g.L >= g.off + 16 + hd4 ? (g.L - g.off - 16 - hd4) / 16 + 1 : 0;
    g.nb = g.nlanes > 0 ? g.off + (g.L - g.off - 16 * g.nlanes) : g.L;
    const uint64_t items = (uint64_t)n * g.rows_tmp * g.nlanes, btotal = (uint64_t)n * g.rows_tmp * g.nb;
    if (items > 0xfffffff0ull || btotal > 0xfffffff0ull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: image too large", fn);
    g.main_items = (unsigned)items;
    g.main_blocks = (unsigned)((items + kThreads - 1) / kThreads);
    g.border_total = (unsigned)btotal;

    VGeom v;
    v.h = h; v.L = g.L; v.pitch = g.pitch; v.rows_tmp = g.rows_tmp; v.K = k;
    v.d_row = d_row; v.d_img = d_img;
    v.nhalf = g.pitch / 8;
    v.nstrips = (v.nhalf + 63) / 64;
    int seg = 4 * RB;                                                // rows per wave item
    if (seg > h) seg = (h + RB - 1) / RB * RB;
    v.seg_rows = seg;
    v.nsegs = (h + seg - 1) / seg;
    const uint64_t vitems = (uint64_t)n * v.nsegs * v.nstrips;
    if (vitems > 0x3fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: image too large", fn);
    v.nitems = (unsigned)vitems;
    TapsV tv{};
    for (int i = 0; i < k; ++i) tv.k[RB - 1 + i] = ky[i];

    const size_t tmp_bytes = (size_t)n * g.rows_tmp * g.pitch * sizeof(uint16_t);
    uint16_t* tmp = nullptr;
    IPX_HIP(hipMallocAsync((void**)&tmp, tmp_bytes, st));
    int rc = IPX_OK;
    if (rmax == 7) launch_h_c<7>(st, src, tmp, g, c, kx, k);
    else launch_h_c<15>(st, src, tmp, g, c, kx, k);
    if (hipGetLastError() != hipSuccess) rc = ipx_fail(IPX_ERR_HIP, "%s: launch of sep_h_kernel failed", fn);
    if (rc == IPX_OK) {
        hipLaunchKernelGGL(sep_v_kernel, dim3((v.nitems + 3) / 4), dim3(kThreads), 0, st, tmp, dst, v, tv);
        if (hipGetLastError() != hipSuccess) rc = ipx_fail(IPX_ERR_HIP, "%s: launch of sep_v_kernel failed", fn);
    }
    hipError_t e = hipFreeAsync(tmp, st);
    if (e != hipSuccess && rc == IPX_OK) rc = ipx_fail(IPX_ERR_HIP, "%s: hipFreeAsync failed: %s", fn, hipGetErrorString(e));
    return rc;
}

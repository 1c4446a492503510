// ipx_sepmfma.hip -- separable fixed-point filter for 9..31 taps on the matrix cores (cv2.GaussianBlur with a large
// kernel, server.py:428; BASELINE.json configs[3]: 15x15 and 31x31 on 8K RGB).
//
// At k = 31 the filter costs 62 multiply-adds per byte: on the vector ALU that is compute-bound (the two-kernel path in
// ipx_sepbig.hip reaches ~5 % of the HBM roofline).  Both passes are banded-Toeplitz matrix products, so they go to the
// int8 MFMA (v_mfma_i32_16x16x64_i8, exact int32 accumulation) and the whole filter becomes one HBM-bound kernel:
//
//   tile = 64 output rows x 128 output byte columns (interleaved channels stay interleaved; taps sit C bytes apart)
//   1. the (64 + 2r) x (128 + 2 HL) source tile goes to LDS as signed bytes (u - 128), REFLECT_101 / halo rows applied
//   2. row pass     H[y][x]  = sum_k S[y][x - HL + k] * Tx[k][x']      A = S rows (ds_read_b128), B = Toeplitz(kx), the
//                                                                     SAME B registers for every tile (shift invariance)
//      h = H + 128*256 is the exact Q8.8 row result; its high and low bytes (again minus 128) are written to two
//      TRANSPOSED LDS planes HT[x][y]
//   3. column pass  O^T[x][y] = sum_k HT[x][y + k] * Ty[k][y']         A = HT columns, B = Toeplitz(ky); one product per
//      plane, out = (256 * (Dhi + 2^15) + (Dlo + 2^15) + 2^15) >> 16
//   4. the result tile is staged in LDS and leaves with full 128-byte row stores.
// Offsetting by 128 is exact because the taps sum to 256.  Taps must fit int8 (<= 127): true for every Gaussian of 9+
// taps; other kernels fall back to ipx_sepbig.hip.  The A/B fragments only rely on A and B sharing the k order inside
// an instruction (sums are order-free in integers).
#include "ipx_common.h"

int ipx_sep_mid(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                const uint16_t* kx, const uint16_t* ky, int k, const uint8_t* top, const uint8_t* bot);

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int TY = 64, TX = 128, kThreads = 256;
constexpr int kRowsS = 96;                       // 6 M-tiles of source rows (64 + 2*15 = 94 used at most)
constexpr int kPitchH = 112;                     // bytes per HT column: >= 96, >= TY - 16 + 64, pitch/4 = 4 mod 8
constexpr int kPitchO = 144;                     // result staging pitch (pitch/4 = 4 mod 8, multiple of 16)

struct MGeom {
    int h, w, L, r, K;
    int64_t s_row, s_img, d_row, d_img;
    int tiles_x, tiles_y;
    unsigned ntiles;
    const uint8_t* top;
    const uint8_t* bot;
};
struct MTaps { uint8_t kx[32]; uint8_t ky[32]; };

__device__ __forceinline__ const uint8_t* src_row(const MGeom& g, const uint8_t* simg, int y)
{
    if (y < 0 && g.top) return g.top + (int64_t)(g.r + y) * g.s_row;
    if (y >= g.h && g.bot) return g.bot + (int64_t)(y - g.h) * g.s_row;
    return simg + (int64_t)reflect101(y, g.h) * g.s_row;
}

__device__ __forceinline__ v4i lds16(const uint8_t* p) { return *reinterpret_cast<const v4i*>(p); }

template <int C, int HL>
__global__ __launch_bounds__(kThreads) void sep_mfma_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                            MGeom g, MTaps taps)
{
    constexpr int KSH = (16 + 2 * HL + 63) / 64;                 // k-steps of the row pass
    constexpr int SW = TX + 2 * HL;                              // source tile width in bytes
    constexpr int PS = HL <= 48 ? 240 : 272;                     // source pitch: >= SW, = 16 or 240 mod 256
    constexpr int NCH = SW / 16;
    __shared__ __attribute__((aligned(16))) uint8_t s_src[kRowsS * PS + 64 * KSH];   // also the result staging tile
    __shared__ __attribute__((aligned(16))) uint8_t s_ht[2][TX * kPitchH];
    __shared__ uint8_t s_taps[64];

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, gq = lane >> 4;
    const unsigned tile = xcd_remap(blockIdx.x, g.ntiles);
    const int tx = (int)(tile % (unsigned)g.tiles_x);
    const int ty = (int)(tile / (unsigned)g.tiles_x % (unsigned)g.tiles_y);
    const unsigned img = tile / ((unsigned)g.tiles_x * (unsigned)g.tiles_y);
    const int x0 = tx * TX, y0 = ty * TY;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    const int r = g.r, K = g.K, RY = TY + 2 * r;

    if (tid < 64) s_taps[tid] = tid < 32 ? taps.kx[tid] : taps.ky[tid - 32];

    // ---- 1. source tile -> LDS (signed bytes) ----------------------------------------------------------------
    for (int idx = tid; idx < RY * NCH; idx += kThreads) {
        const int row = idx / NCH, ch = idx - row * NCH;
        int ys = y0 - r + row;
        if (ys > g.h + r - 1) ys = g.h + r - 1;                     // rows below the last one anybody needs: clamp
        const uint8_t* rp = src_row(g, simg, ys);
        const int gx = x0 - HL + 16 * ch;
        u32x4 v;
        if (gx >= 0 && gx + 16 <= g.L) {
            v = ld16(rp + gx);
        } else {                                                  // image border: REFLECT_101 in pixel space
            uint32_t wds[4] = {0, 0, 0, 0};
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const int col = gx + b;
                int px = col >= 0 ? col / C : -((-col + C - 1) / C);
                const int chn = col - px * C;
                if (px >= -r && px <= g.w - 1 + r) {              // columns further out are never used
                    px = reflect101(px, g.w);
                    wds[b >> 2] |= (uint32_t)rp[px * C + chn] << (8 * (b & 3));
                }
            }
            v = (u32x4){wds[0], wds[1], wds[2], wds[3]};
        }
        v ^= (u32x4)(0x80808080u);
        *reinterpret_cast<u32x4*>(s_src + row * PS + 16 * ch) = v;
    }
    __syncthreads();

    // ---- Toeplitz operands (registers, the same for every tile) ----------------------------------------------
    v4i bh[KSH], bv;
    {
        const int n = m;                                          // B column = lane & 15
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
            uint32_t wds[4] = {0, 0, 0, 0};
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int k = 64 * ks + 16 * gq + jj;
                const int d = k - HL - n + r * C;                 // = j * C for tap j
                uint32_t t = 0;
                if (d >= 0 && d % C == 0 && d / C < K) t = s_taps[d / C];
                wds[jj >> 2] |= t << (8 * (jj & 3));
            }
            bh[ks] = (v4i){(int)wds[0], (int)wds[1], (int)wds[2], (int)wds[3]};
        }
        uint32_t wds[4] = {0, 0, 0, 0};
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int i = 16 * gq + jj - n;
            uint32_t t = 0;
            if (i >= 0 && i < K) t = s_taps[32 + i];
            wds[jj >> 2] |= t << (8 * (jj & 3));
        }
        bv = (v4i){(int)wds[0], (int)wds[1], (int)wds[2], (int)wds[3]};
    }

    // ---- 2. row pass: wave w owns column tiles t = w and w + 4 of every 16-row tile --------------------------
    const int MT = (RY + 15) / 16;
    for (int mt = 0; mt < MT; ++mt) {
        const uint8_t* arow = s_src + (16 * mt + m) * PS + 16 * gq + 16 * wave;
        v4i ch[KSH + 1];
#pragma unroll
        for (int q = 0; q <= KSH; ++q) ch[q] = lds16(arow + 64 * q);   // tile w uses ch[0..KSH), tile w+4 ch[1..KSH]
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            v4i acc = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks)
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(ch[ks + half], bh[ks], acc, 0, 0, 0);
            // acc[i] = row 16 mt + 4 gq + i, column 16 t + m; Q8.8 value h = acc + 2^15
            const uint32_t h0 = (uint32_t)acc[0] + 32768u, h1 = (uint32_t)acc[1] + 32768u;
            const uint32_t h2 = (uint32_t)acc[2] + 32768u, h3 = (uint32_t)acc[3] + 32768u;
            const uint32_t hi = (__builtin_amdgcn_perm(h1, h0, 0x0c0c0501u) | __builtin_amdgcn_perm(h3, h2, 0x05010c0cu)) ^ 0x80808080u;
            const uint32_t lo = (__builtin_amdgcn_perm(h1, h0, 0x0c0c0400u) | __builtin_amdgcn_perm(h3, h2, 0x04000c0cu)) ^ 0x80808080u;
            const int col = 16 * (wave + 4 * half) + m;
            *reinterpret_cast<uint32_t*>(&s_ht[0][col * kPitchH + 16 * mt + 4 * gq]) = hi;
            *reinterpret_cast<uint32_t*>(&s_ht[1][col * kPitchH + 16 * mt + 4 * gq]) = lo;
        }
    }
    __syncthreads();

    // ---- 3. column pass: wave w owns column tiles 2w, 2w+1; result -> staging tile (aliases the source tile) ---
    uint8_t* s_out = s_src;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
        const int mt = 2 * wave + mh;
        const uint8_t* ahi = &s_ht[0][(16 * mt + m) * kPitchH + 16 * gq];
        const uint8_t* alo = &s_ht[1][(16 * mt + m) * kPitchH + 16 * gq];
#pragma unroll
        for (int nt = 0; nt < TY / 16; ++nt) {
            const v4i z = {0, 0, 0, 0};
            const v4i dhi = __builtin_amdgcn_mfma_i32_16x16x64_i8(lds16(ahi + 16 * nt), bv, z, 0, 0, 0);
            const v4i dlo = __builtin_amdgcn_mfma_i32_16x16x64_i8(lds16(alo + 16 * nt), bv, z, 0, 0, 0);
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (uint32_t)(dhi[i] * 256 + dlo[i] + (32768 * 256 + 32768 + 32768));
            // o[i] bits 16..23 = output byte of column 16 mt + 4 gq + i, row 16 nt + m
            const uint32_t pk = __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0602u) | __builtin_amdgcn_perm(o[3], o[2], 0x06020c0cu);
            *reinterpret_cast<uint32_t*>(s_out + (16 * nt + m) * kPitchO + 16 * mt + 4 * gq) = pk;
        }
    }
    __syncthreads();

    // ---- 4. staging tile -> global, 16 bytes per lane, whole 128-byte rows -----------------------------------
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    for (int idx = tid; idx < TY * (TX / 16); idx += kThreads) {
        const int row = idx / (TX / 16), cc = idx % (TX / 16);
        const int y = y0 + row, x = x0 + 16 * cc;
        if (y >= g.h || x >= g.L) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(s_out + row * kPitchO + 16 * cc);
        uint8_t* dp = dimg + (int64_t)y * g.d_row + x;
        if (x + 16 <= g.L) {
            st16_nt(dp, v);
        } else {
            for (int b = 0; x + b < g.L; ++b) dp[b] = (uint8_t)(v[b >> 2] >> (8 * (b & 3)));
        }
    }
}

template <int C>
void launch_c(hipStream_t st, const uint8_t* src, uint8_t* dst, const MGeom& g, const MTaps& t)
{
    const int hl = (g.r * C + 15) & ~15;
    const dim3 grid(g.ntiles), block(kThreads);
    switch (hl) {
        case 16: hipLaunchKernelGGL((sep_mfma_kernel<C, 16>), grid, block, 0, st, src, dst, g, t); break;
        case 32: hipLaunchKernelGGL((sep_mfma_kernel<C, 32>), grid, block, 0, st, src, dst, g, t); break;
        case 48: hipLaunchKernelGGL((sep_mfma_kernel<C, 48>), grid, block, 0, st, src, dst, g, t); break;
        default: hipLaunchKernelGGL((sep_mfma_kernel<C, 64>), grid, block, 0, st, src, dst, g, t); break;
    }
}

}  // namespace

// Called by ipx_sep_filter_q8_u8 for 9 <= k <= 31 (arguments validated there).
int ipx_sep_mfma(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                 int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                 const uint16_t* kx, const uint16_t* ky, int k, const uint8_t* top, const uint8_t* bot)
{
    const char* fn = "ipx_sep_filter_q8_u8";
    bool fits = true;
    for (int i = 0; i < k; ++i) fits = fits && kx[i] <= 127 && ky[i] <= 127;
    if (!fits) return ipx_sep_mid(st, src, dst, n, h, w, c, s_row, s_img, d_row, d_img, kx, ky, k, top, bot);
    MGeom g;
    g.h = h; g.w = w; g.L = w * c; g.r = k / 2; g.K = k;
    g.s_row = s_row; g.s_img = s_img; g.d_row = d_row; g.d_img = d_img; g.top = top; g.bot = bot;
    g.tiles_x = (g.L + TX - 1) / TX;
    g.tiles_y = (h + TY - 1) / TY;
    const uint64_t nt = (uint64_t)n * g.tiles_x * g.tiles_y;
    if (nt > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: image too large", fn);
    g.ntiles = (unsigned)nt;
    MTaps t{};
    for (int i = 0; i < k; ++i) { t.kx[i] = (uint8_t)kx[i]; t.ky[i] = (uint8_t)ky[i]; }
    switch (c) {
        case 1: launch_c<1>(st, src, dst, g, t); break;
        case 2: launch_c<2>(st, src, dst, g, t); break;
        case 3: launch_c<3>(st, src, dst, g, t); break;
        default: launch_c<4>(st, src, dst, g, t); break;
    }
    IPX_LAUNCH_CHECK("sep_mfma_kernel");
    return IPX_OK;
}

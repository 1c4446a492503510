// ipx_medianrow.hip -- cv2.medianBlur for 13 <= k <= 31 (server.py:215, :430).
//
// A lane per image ROW.  A wave owns 64 consecutive rows and slides along them; every lane keeps the histogram of ITS k x k
// window in LDS, laid out [bin][lane] so that lane l always touches bank l % 32 whatever the pixel values are (a constant
// image costs the same as noise), and moving one pixel to the right is k `ds_add_u32` for the column that enters and k
// `ds_sub_u32` for the one that leaves (Huang's update: O(k) per pixel, but each is ONE instruction for 64 pixels, with no
// dependency between them).  A second, 16-bin histogram of the high nibbles (Perreault-Hebert's coarse level) bounds the rank
// search at 16 + 16 bins.  Counters are packed: 13, 15 taps (<= 225 per bin) four 8-bit counters per dword, 17..31 taps (<= 961)
// two 16-bit counters; a value's dword is its low 6 / 7 bits and its part the bits above, so the fine and the coarse
// increment are the same number.
//
// The sliding-column-histogram kernel this replaces (median_hist_kernel, ipx_rank.hip) spread the 256 bins of ONE window
// over the wave and paid a wave-wide prefix sum per pixel: ~40 instructions per pixel at any k, 19.7 ms per 24 x 4K at k = 31.
//
// Pixels reach the lanes through a ring of 64 columns in LDS, stored column-major ([column][tile row], the tile being the
// wave's 64 rows plus r above and below, replicated at the image border as cv2 does): a lane's k window rows of one column
// are k consecutive bytes, read as aligned dwords and shifted into place.  The ring is refilled 32 columns at a time from
// byte loads issued one chunk ahead.
#include "ipx_common.h"
#include <stdlib.h>

namespace {

struct RowMedGeom {
    int h, w;
    int64_t s_row, s_img, d_row, d_img;
    int bands, segs, span;                       // 64-row bands per image, column segments per band, columns per segment
};

template <int K>
struct RowMedCfg {
    static constexpr bool kWide = K * K > 255;
    static constexpr int R = K / 2;
    static constexpr int kTileRows = 64 + 2 * R;
    static constexpr int kRowP = (kTileRows + 3) & ~3;                 // bytes per ring column
    static constexpr int kFineRows = kWide ? 128 : 64;                 // dwords of fine counters per lane
    static constexpr int kCoarseRows = kWide ? 8 : 4;
    static constexpr int kBits = kWide ? 16 : 8;                       // counter width
    static constexpr int kShift = kWide ? 7 : 6;                       // value >> kShift = the part of the dword
    static constexpr int kHistBytes = (kFineRows + kCoarseRows) * 256;
    static constexpr int kOut = (K + 3) / 4;                           // dwords that hold a lane's k window bytes of one column
    static constexpr int kWords = kOut + 1;                            // aligned dwords read to form them at any byte offset
    static constexpr int kOver = 60 + 4 * kWords > kRowP ? 60 + 4 * kWords - kRowP : 0;   // lane 63's reads past the last column
    static constexpr int kLdsBytes = kHistBytes + 64 * kRowP + kOver;  // 40960 at 31 taps (4 waves per CU), <= 22528 at 13, 15
};

template <int K>
__global__ __launch_bounds__(64) void median_row_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, RowMedGeom g)
{
    using C = RowMedCfg<K>;
    constexpr int R = C::R;
    extern __shared__ uint32_t lds[];
    uint8_t* const tile = (uint8_t*)lds + C::kHistBytes;
    const int lane = (int)threadIdx.x;
    const unsigned blk = blockIdx.x;
    const int seg = (int)(blk % (unsigned)g.segs);
    const int band = (int)(blk / (unsigned)g.segs % (unsigned)g.bands);
    const unsigned img = blk / ((unsigned)g.segs * (unsigned)g.bands);
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* dimg = dst + (int64_t)img * g.d_img;
    const int y0 = band * 64, x_start = seg * g.span;
    const int x_end = x_start + g.span < g.w ? x_start + g.span : g.w;
    const int nsteps = x_end - x_start + 2 * R;                       // stream position s adds image column x_start - R + s
    const int need = K * K / 2 + 1;

    // this lane's own histograms: dword D of lane l at lds[D * 64 + l]
    uint32_t* const fine = lds + lane;
    uint32_t* const coarse = lds + C::kFineRows * 64 + lane;
#pragma unroll 8
    for (int i = 0; i < C::kFineRows + C::kCoarseRows; ++i) lds[i * 64 + lane] = 0u;

    // the (up to) two tile rows this lane stages: tile row t is image row clamp(y0 - R + t)
    const uint8_t* rowp[2];
    bool stage[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int t = lane + 64 * p;
        stage[p] = t < C::kTileRows;
        int y = y0 - R + t;
        y = y < 0 ? 0 : y >= g.h ? g.h - 1 : y;
        rowp[p] = simg + (int64_t)y * g.s_row;
    }
    uint8_t pf[2][32];
    auto fetch = [&](int q) {                                          // stream columns 32 q .. 32 q + 31 into registers
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            int c = x_start - R + 32 * q + i;
            c = c < 0 ? 0 : c >= g.w ? g.w - 1 : c;                    // BORDER_REPLICATE
#pragma unroll
            for (int p = 0; p < 2; ++p)
                if (p == 0 || stage[1]) pf[p][i] = rowp[p][c];
        }
    };
    auto deposit = [&](int q) {                                        // ... and into ring half q & 1
        uint8_t* base = tile + (32 * (q & 1)) * C::kRowP + lane;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            base[i * C::kRowP] = pf[0][i];
            if (stage[1]) base[i * C::kRowP + 64] = pf[1][i];
        }
    };

    // k bytes of ring column `col` from tile row `lane` on: kWords aligned dwords shifted so that byte j is the window's row j
    auto column = [&](int col, uint32_t (&wv)[C::kOut]) {
        const uint32_t* p = (const uint32_t*)(tile + col * C::kRowP + (lane & ~3));
        uint32_t d[C::kWords];
#pragma unroll
        for (int i = 0; i < C::kWords; ++i) d[i] = p[i];
        const uint32_t sh = (uint32_t)(lane & 3) * 8u;
#pragma unroll
        for (int i = 0; i + 1 < C::kWords; ++i) wv[i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], sh);
    };
    // One update per window row: bit fields straight out of the packed column dwords (v_bfe_u32), address = field << 8 | lane
    // (v_lshl_add_u32), increment = part * 0xffff + 1 (v_mad_u32_u24): 6 vector instructions and 2 LDS atomics per pixel
    // value with 16-bit counters, 7 with 8-bit ones.  Written as instructions because the compiler turns the bit-field
    // extracts back into shift / and / compare / select sequences (9 per value).
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    // (the LDS address of the lane's column of counters, as a number: the field << 8 is added to it by the same instruction,
    // and the atomics take the sum as it is -- through a generic pointer every address would cost one more add)
    const uint32_t lane4 = (uint32_t)(uintptr_t)(lds_u32*)lds + (uint32_t)lane * 4u;
    const uint32_t k_ffff = 0xffffu;
    auto bump = [&](const uint32_t (&wv)[C::kOut], bool add) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t word = wv[j >> 2];
            constexpr int kCoarseBits = C::kWide ? 3 : 2;
            uint32_t lo, part, cb, fo, co, inc;
            asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(lo) : "v"(word), "n"(8 * (j & 3)), "n"(C::kShift));
            asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(part) : "v"(word), "n"(8 * (j & 3) + C::kShift), "n"(8 - C::kShift));
            asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(cb) : "v"(word), "n"(8 * (j & 3) + 4), "n"(kCoarseBits));
            asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(fo) : "v"(lo), "v"(lane4));
            asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(co) : "v"(cb), "v"(lane4));
            if (C::kWide) {
                asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(inc) : "v"(part), "v"(k_ffff));
            } else {
                uint32_t p8;
                asm("v_lshlrev_b32 %0, 3, %1" : "=v"(p8) : "v"(part));
                asm("v_lshlrev_b32 %0, %1, 1" : "=v"(inc) : "v"(p8));
            }
            lds_u32* f = (lds_u32*)(uintptr_t)fo;
            lds_u32* c = (lds_u32*)(uintptr_t)co + C::kFineRows * 64;
#ifdef IPX_MEDIAN_AB     // compile-time timing experiments (-DIPX_MEDIAN_AB=n, wrong pixels): 2 no coarse atomics, 16 no fine atomics,
                         // 32 atomics on one fixed address (no address arithmetic), 64 address arithmetic only.  24 x 4K, k = 31, 7.2 ms:
                         // atomics 2.5 ms + their address arithmetic 3.3 ms + column reads, rank search and stores 2.0 ms; the
                         // three ADD UP (a wave's LDS and vector instructions issue in order, one wave per SIMD at 17+ taps)
            constexpr bool with_coarse = !(IPX_MEDIAN_AB & 2), with_fine = !(IPX_MEDIAN_AB & 16);
            if (IPX_MEDIAN_AB & 32) { f = (lds_u32*)(uintptr_t)lane4; c = f + 64; inc = 1u; }
            if (IPX_MEDIAN_AB & 64) { asm volatile("" :: "v"(fo), "v"(co), "v"(inc)); continue; }
#else
            constexpr bool with_coarse = true, with_fine = true;
#endif
            if (add) {
                if (with_fine) __hip_atomic_fetch_add(f, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (with_coarse) __hip_atomic_fetch_add(c, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                if (with_fine) __hip_atomic_fetch_sub(f, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (with_coarse) __hip_atomic_fetch_sub(c, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    // smallest value whose cumulative count reaches `need`: the part (values 64 p .. / 128 p ..), then the coarse bin inside
    // it, then the value inside the coarse bin.  Prefix sums run on the packed dwords (no counter of a prefix exceeds k*k, so
    // nothing carries between the parts); only the wanted part is pulled out of each.
    auto rank = [&]() -> uint32_t {
        uint32_t cp[C::kCoarseRows];
#pragma unroll
        for (int i = 0; i < C::kCoarseRows; ++i) cp[i] = coarse[i << 6];
#pragma unroll
        for (int i = 1; i < C::kCoarseRows; ++i) cp[i] += cp[i - 1];
        const uint32_t tot = cp[C::kCoarseRows - 1];
        constexpr uint32_t kMask = (1u << C::kBits) - 1u;
        uint32_t part = 0, below = 0, run = 0;
#pragma unroll
        for (int p = 0; p + 1 < 32 / C::kBits; ++p) {
            run += (tot >> (p * C::kBits)) & kMask;
            const bool lt = run < (uint32_t)need;
            part += lt ? 1u : 0u;
            below = lt ? run : below;
        }
        const uint32_t psh = part * (uint32_t)C::kBits;
        uint32_t want = (uint32_t)need - below;                         // rank inside the part
        uint32_t idx = 0, under = 0;
#pragma unroll
        for (int i = 0; i + 1 < C::kCoarseRows; ++i) {
            const uint32_t e = __builtin_amdgcn_ubfe(cp[i], psh, C::kBits);
            const bool lt = e < want;
            idx += lt ? 1u : 0u;
            under = lt ? e : under;
        }
        want -= under;                                                  // rank inside the coarse bin
        // coarse bin = part * kCoarseRows + idx; its 16 values live in fine dwords 16 idx .. 16 idx + 15, same part
        const uint32_t* f = fine + (idx << 10);
        uint32_t fp[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) fp[t] = f[t << 6];
#pragma unroll
        for (int t = 1; t < 15; ++t) fp[t] += fp[t - 1];
        uint32_t low = 0;
#pragma unroll
        for (int t = 0; t < 15; ++t) low += __builtin_amdgcn_ubfe(fp[t], psh, C::kBits) < want ? 1u : 0u;
        return (part << C::kShift) + (idx << 4) + low;
    };

    const int nq = (nsteps + 31) / 32;
    fetch(0);
    deposit(0);
    const int y = y0 + lane;
    uint8_t* drow = dimg + (int64_t)(y < g.h ? y : g.h - 1) * g.d_row + x_start - 2 * R;
    const bool live = y < g.h;
    for (int q = 0; q < nq; ++q) {
        if (q + 1 < nq) fetch(q + 1);                                   // in flight during this chunk's 32 steps
        const int s_end = 32 * q + 32 < nsteps ? 32 * q + 32 : nsteps;
        for (int s = 32 * q; s < s_end; ++s) {
            uint32_t wv[C::kOut];
            column(s & 63, wv);
            bump(wv, true);
            if (s >= K) {
                column((s - K) & 63, wv);
                bump(wv, false);
            }
            if (s >= 2 * R) {
                const uint32_t m = rank();
                if (live) drow[s] = (uint8_t)m;
            }
        }
        if (q + 1 < nq) deposit(q + 1);
    }
}

template <int K>
void launch_row_median(hipStream_t st, const uint8_t* src, uint8_t* dst, const RowMedGeom& g, unsigned blocks)
{
    hipLaunchKernelGGL(median_row_kernel<K>, dim3(blocks), dim3(64), (size_t)RowMedCfg<K>::kLdsBytes, st, src, dst, g);
}

}  // namespace

// Called by ipx_median_blur_u8 for 13 <= k <= 31 (arguments already checked there): returns 1 if it launched, < 0 on error.
int ipx_median_rows(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                    int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img, int k)
{
    RowMedGeom g;
    g.h = h; g.w = w; g.s_row = s_row; g.s_img = s_img; g.d_row = d_row; g.d_img = d_img;
    g.bands = (h + 63) / 64;
    // column segments: a wave walks its segment serially and pays 2r extra steps to fill its first window, so segments are
    // long; but there must be several waves per SIMD-slot over the launch for the tail to be short (4 waves per CU with
    // 16-bit counters, 7 with 8-bit ones)
    const int64_t slots = k > 15 ? 1024 : 1792;
    const int64_t rows = (int64_t)n * g.bands;
    int64_t segs = (4 * slots + rows - 1) / rows;
    const int64_t max_segs = w / 128 > 1 ? w / 128 : 1;
    if (segs > max_segs) segs = max_segs;
    if (segs < 1) segs = 1;
#ifdef IPX_DEV
    if (const char* e = getenv("IPX_MEDIAN_SEGS")) { if (atoi(e) > 0) segs = atoi(e); }
#endif
    g.span = (int)((w + segs - 1) / segs);
    g.segs = (w + g.span - 1) / g.span;
    const uint64_t nb = (uint64_t)rows * (uint64_t)g.segs;
    if (nb > 0x7fffffffull) return ipx_fail(IPX_ERR_UNSUPPORTED, "ipx_median_blur_u8: too many tiles");
    const unsigned blocks = (unsigned)nb;
    switch (k) {
    case 13: launch_row_median<13>(st, src, dst, g, blocks); break;
    case 15: launch_row_median<15>(st, src, dst, g, blocks); break;
    case 17: launch_row_median<17>(st, src, dst, g, blocks); break;
    case 19: launch_row_median<19>(st, src, dst, g, blocks); break;
    case 21: launch_row_median<21>(st, src, dst, g, blocks); break;
    case 23: launch_row_median<23>(st, src, dst, g, blocks); break;
    case 25: launch_row_median<25>(st, src, dst, g, blocks); break;
    case 27: launch_row_median<27>(st, src, dst, g, blocks); break;
    case 29: launch_row_median<29>(st, src, dst, g, blocks); break;
    case 31: launch_row_median<31>(st, src, dst, g, blocks); break;
    default: return ipx_fail(IPX_ERR_INVALID, "ipx_median_rows: k=%d outside 13..31", k);
    }
    if (hipGetLastError() != hipSuccess) return ipx_fail(IPX_ERR_HIP, "launch of median_row_kernel<%d> failed", k);
    return 1;
}

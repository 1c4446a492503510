// ipx_sepmfma.hip -- separable fixed-point filter for 9..31 taps on the matrix cores (cv2.GaussianBlur with a large
// kernel, server.py:428; BASELINE.json configs[3]: 15x15 and 31x31 on 8K RGB).
//
// At k = 31 the filter costs 62 multiply-adds per byte: on the vector ALU that is compute-bound (the two-kernel path in
// ipx_sepbig.hip reaches ~5 % of the HBM roofline).  Both passes are banded-Toeplitz matrix products, so they go to the
// int8 MFMA (v_mfma_i32_16x16x64_i8, exact int32 accumulation) and the whole filter becomes one HBM-bound kernel:
//
//   tile = 64 output rows x 128 output byte columns, one 512-thread block (8 waves, one per 16-column slice) (interleaved channels stay interleaved; taps sit C bytes apart)
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
#include <stdlib.h>
#include <stdio.h>

int ipx_sep_mid(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                const uint16_t* kx, const uint16_t* ky, int k, const uint8_t* top, const uint8_t* bot);
int ipx_sep_wave(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                 int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                 const uint16_t* kx, const uint16_t* ky, int k, const uint8_t* top, const uint8_t* bot);

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int TY = 64, TX = 128, kThreads = 512;       // 8 waves per tile: 3 resident blocks = 24 waves per CU
constexpr int kRowsS = 96;                       // 6 M-tiles of source rows (64 + 2*15 = 94 exist at most)
constexpr int kPitchHRot = 128, kPitchHPlain = 96;  // bytes per HT column.
// Plain (96 = 32 mod 64): the column pass's ds_read_b128 are conflict-free in every lane group of MI355X_MICROARCH.md section
// LDS, but the row pass's ds_write_b32 are 4-way (banks (24 m + gq) mod 32).  Rotated (128 = 8 chunks of 16 rows, 6 of them
// hold the 96 row-pass rows): column 16 w + m keeps its chunk c at position (c + rot(m)) mod 8, rot(m) = (m & 3) + 4 (m >> 3);
// reads stay conflict-free, stores become 2-way (free for stores), and the hi / lo stores of a lane (planes 64 x 256 B apart)
// merge into one ds_write2st64_b32 (scripts/dev/ldsbanks.py enumerates pitches and rotations; none exists for a pitch below
// 128).  Measured: 3-5 % (9 taps on 2 x 8K BGR 116-120 us against 121-123, 15 taps ~140 against ~150) -- LDS cycles are not what
// bounds this kernel.  The rotated planes cost 8 KiB more, and LDS is handed out in granules of 1280 bytes: three blocks per CU
// need <= 42 granules = 53,760 bytes each (53,824 already run two per CU: 31 taps 160 -> 186 us).  So the rotation is used where
// the source tile is small enough (k <= 17: 80 rows); 19-31 taps (94 rows, 53,824 bytes with it) and 64-byte halos stay plain.
// The column pass's last k-chunk (rows 96..111, zero taps) reads finite garbage: the next column / the bytes behind the plane
// (plain), or a never-written chunk of the same column (rotated).
constexpr int kPitchO = 144;                     // result staging pitch (pitch/4 = 4 mod 8, multiple of 16)

struct MGeom {
    int h, w, L, r, K;
    int64_t s_row, s_img, d_row, d_img;
    int tiles_x, tiles_y;
    int pw, ph, patches_x, patches_y;                // an XCD's blocks work on a pw x ph patch of tiles at a time
    unsigned npatches;
    const uint8_t* top;
    const uint8_t* bot;
    int fast32;                                      // row strides small enough for 32-bit in-tile byte offsets
#ifdef IPX_DEV
    int ablate;                                      // dev build: skip phases (timing only, results wrong)
    long long* prof;                                 // dev build: phase time stamps of block 0 (IPX_MFMA_PROF), else null
#endif
};
// development builds (make EXTRA=-DIPX_DEV) can skip phases and stamp them; the shipped library has neither
#ifdef IPX_DEV
#define IPX_PHASE_ON(g, bit) (!((g).ablate & (bit)))
#else
#define IPX_PHASE_ON(g, bit) true
#endif
struct MTaps { uint8_t kx[32]; uint8_t ky[32]; };

__device__ __forceinline__ const uint8_t* src_row(const MGeom& g, const uint8_t* simg, int y)
{
    return halo_row(simg, g.top, g.bot, y, g.h, g.r, g.s_row);
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's outstanding GLOBAL loads
// (vmcnt(0)), which would serialise the next tile's prefetch behind every barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ v4i lds16(const uint8_t* p) { return *reinterpret_cast<const v4i*>(p); }

template <int C, int HL, int NMT>                               // NMT: 16-row tiles of the row pass (5 for k <= 17, else 6)
__global__ __launch_bounds__(kThreads, 6) void sep_mfma_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                               MGeom g, MTaps taps)
{
    constexpr int KSH = (16 + 2 * HL + 63) / 64;                 // k-steps of the row pass
    constexpr int SW = TX + 2 * HL;                              // source tile width in bytes
    constexpr int PS = HL <= 48 ? 224 : 288;                     // source pitch: >= SW and = 32 mod 64: conflict-free b128 reads
    constexpr int NCH = SW / 16;
    constexpr int kPitchH = (HL > 48 || NMT != 5) ? kPitchHPlain : kPitchHRot;
    constexpr bool ROT = kPitchH == kPitchHRot;
    constexpr int NLD = kRowsS / 32;                             // source rows per thread: rows (tid >> 4) + 32 i
    // one LDS block: source tile (also the result staging tile) | the two transposed planes | taps.  Only TY + 30 = 94
    // source rows exist; the row pass's last 16-row tile and its k-step padding read on into the planes (times zero taps)
    constexpr int kSrcBytes = (NMT == 5 ? 80 : TY + 30) * PS;    // k <= 17: 64 + 2 * 8 = 80 rows at most
    __shared__ __attribute__((aligned(16))) uint8_t s_all[kSrcBytes + 2 * TX * kPitchH + 64 + 16];
    uint8_t* const s_src = s_all;
    uint8_t* const s_ht0 = s_all + kSrcBytes;
    uint8_t* const s_ht1 = s_ht0 + TX * kPitchH;
    uint8_t* const s_taps = s_ht1 + TX * kPitchH;

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, gq = lane >> 4;
    const int r = g.r, K = g.K, RY = TY + 2 * r;
    const int lrow = tid >> 4, lch = tid & 15;                   // chunk column lch < NCH (NCH <= 16)

    if (tid < 64) s_taps[tid] = tid < 32 ? taps.kx[tid] : taps.ky[tid - 32];

    // Blocks b, b + 8, b + 16, ... share an XCD (round-robin dispatch), so block b = 8 q + x is tile q of the pw x ph
    // patch that XCD x works on; the XCD walks patches x, x + 8, x + 16, ...  Neighbouring tiles of a patch are in
    // flight together and share their halos through that XCD's L2.  The next tile of a block is prefetched into
    // registers while the current one is filtered.  Patch coordinates advance incrementally (no divisions per tile).
    struct Pos { int ppx, ppy; unsigned img, P; };
    const int q = (int)(blockIdx.x >> 3), qx = q % g.pw, qy = q / g.pw;
    const int ddx = 8 % g.patches_x, ddy = 8 / g.patches_x % g.patches_y;
    const unsigned ddi = 8u / ((unsigned)g.patches_x * (unsigned)g.patches_y);
    auto advance = [&](Pos p) {
        p.P += 8;
        p.ppx += ddx; if (p.ppx >= g.patches_x) { p.ppx -= g.patches_x; ++p.ppy; }
        p.ppy += ddy; if (p.ppy >= g.patches_y) { p.ppy -= g.patches_y; ++p.img; }
        p.img += ddi;
        return p;
    };
    auto tile_x = [&](const Pos& p) { return p.ppx * g.pw + qx; };
    auto tile_y = [&](const Pos& p) { return p.ppy * g.ph + qy; };
    auto valid = [&](const Pos& p) { return p.P < g.npatches && tile_x(p) < g.tiles_x && tile_y(p) < g.tiles_y; };
    // per-lane byte offsets of its NLD source chunks inside an interior tile (idle lanes re-load a clamped address,
    // so that no load sits behind a branch and all NLD stay in flight)
    unsigned loff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i)
        loff[i] = (unsigned)((lrow + 32 * i < RY ? lrow + 32 * i : RY - 1) * (int)g.s_row + 16 * (lch < NCH ? lch : NCH - 1));

    auto fetch = [&](const Pos& p, u32x4 (&reg)[NLD]) {
        const uint8_t* simg = src + (int64_t)p.img * g.s_img;
        const int xs0 = tile_x(p) * TX - HL, ys0 = tile_y(p) * TY - r;
        if (g.fast32 && xs0 >= 0 && xs0 + SW <= g.L && ys0 >= 0 && ys0 + RY <= g.h) {   // interior tile (block-uniform)
            const uint8_t* base = simg + (int64_t)ys0 * g.s_row + xs0;
#pragma unroll
            for (int i = 0; i < NLD; ++i) reg[i] = ld16(base + loff[i]);
            return;
        }
        // border tiles (a few per cent of all): still branch-free per row so that the loads overlap.  Rows come from
        // src_row (REFLECT_101 or the caller's halo rows); a tile that crosses the left/right image edge gathers its
        // bytes one by one with REFLECT_101 in pixel space (16 independent byte loads per chunk).
        const int gx = xs0 + 16 * (lch < NCH ? lch : NCH - 1);
        const bool xin = xs0 >= 0 && xs0 + SW <= g.L;            // block-uniform
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int row = lrow + 32 * i < RY ? lrow + 32 * i : RY - 1;
            int ys = ys0 + row;
            if (ys > g.h + r - 1) ys = g.h + r - 1;                 // rows below the last one anybody needs: clamp
            const uint8_t* rp = src_row(g, simg, ys);
            if (xin) {
                reg[i] = ld16(rp + gx);
            } else {                                              // 4 x 4 byte loads; kept apart so that this rare path
                uint32_t wds[4];                                  // does not raise the kernel's register count
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) {
                    uint32_t word = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int col = gx + 4 * wd + b;
                        int px = col >= 0 ? col / C : -((-col + C - 1) / C);
                        const int chn = col - px * C;
                        px = px < -r ? -r : px > g.w - 1 + r ? g.w - 1 + r : px;   // further out: never used
                        word |= (uint32_t)rp[reflect101(px, g.w) * C + chn] << (8 * b);
                    }
                    wds[wd] = word;
                    __builtin_amdgcn_sched_barrier(0);
                }
                reg[i] = (u32x4){wds[0], wds[1], wds[2], wds[3]};
            }
        }
    };
    Pos pa;
    pa.P = blockIdx.x & 7u;
    pa.ppx = (int)(pa.P % (unsigned)g.patches_x);
    pa.ppy = (int)(pa.P / (unsigned)g.patches_x % (unsigned)g.patches_y);
    pa.img = pa.P / ((unsigned)g.patches_x * (unsigned)g.patches_y);
    u32x4 bufa[NLD];
    if (valid(pa)) fetch(pa, bufa);
    __syncthreads();                                              // s_taps

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

    // per-lane LDS addresses (tile independent)
    uint8_t* const st_src = s_src + lrow * PS + 16 * lch;                       // source staging: + 32 i PS
    const uint8_t* const a_h = s_src + m * PS + 16 * gq + 16 * wave;            // row pass A: + 16 mt PS + 64 q
    const int rot = ROT ? (m & 3) + 4 * (m >> 3) : 0;                           // chunk rotation of column 16 w + m (see kPitchHRot)
    uint8_t* const w_hi = s_ht0 + (16 * wave + m) * kPitchH + 4 * gq;           // row pass result: + 16 ((mt + rot) & 7); lo plane:
    constexpr int kPlane = TX * kPitchH;                                        // + kPlane (= 64 x 256 B: one ds_write2st64_b32)
    const uint8_t* const a_vhi = s_ht0 + (16 * wave + m) * kPitchH;             // column pass A: + 16 ((nt + gq + rot) & 7)
    const int rrot = gq + rot;
    uint8_t* const s_out = s_src;
    uint8_t* const w_out = s_out + m * kPitchO + 16 * wave + 4 * gq;            // + 16 nt kPitchO
    const int orow = tid >> 3, occ = tid & 7;                                   // output chunk: row orow (0..63)
    const uint8_t* const r_out = s_out + orow * kPitchO + 16 * occ;
    const unsigned ooff = (unsigned)(orow * (int)g.d_row + 16 * occ);

#ifdef IPX_DEV
    int pslot = 0;
    auto stamp = [&]() { if (g.prof && blockIdx.x == 0 && tid == 0 && pslot < 64) g.prof[pslot++] = (long long)__builtin_readcyclecounter(); };
#else
    auto stamp = [] {};
#endif
    stamp();
    auto do_tile = [&](const Pos& cur, const Pos& fut, u32x4 (&nxt)[NLD]) {
    const bool live = valid(cur);                                 // block-uniform
    // ---- 1. source tile (fetched during the previous tile) -> LDS as signed bytes; refill the registers --------
    if (live) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (lch < NCH && lrow + 32 * i < RY) *reinterpret_cast<u32x4*>(st_src + 32 * i * PS) = nxt[i] ^ (u32x4)(0x80808080u);
        lds_barrier();
    }
    stamp();
    if (valid(fut) && IPX_PHASE_ON(g, 1)) fetch(fut, nxt);
    stamp();
    if (!live) return;

    // ---- 2. row pass: wave w owns column tile w (16 output columns) of every 16-row tile ----------------------
    // NMT whole 16-row tiles, also when fewer source rows exist (rows >= RY hold stale bytes that only meet zero taps):
    // no branch inside the unrolled loop, so the LDS reads of later tiles overlap the MFMAs and epilogues of earlier ones
    if (IPX_PHASE_ON(g, 2))
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        v4i acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks)
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(lds16(a_h + 16 * mt * PS + 64 * ks), bh[ks], acc, 0, 0, 0);
        // acc[i] = row 16 mt + 4 gq + i, column 16 w + m; Q8.8 value h = acc + 2^15
        // (h >> 8) - 128 is byte 1 of acc itself ((acc + 2^15) ^ 2^15 == acc in 16 bits); (h & 255) - 128 = byte 0 ^ 0x80
        const uint32_t h0 = (uint32_t)acc[0], h1 = (uint32_t)acc[1], h2 = (uint32_t)acc[2], h3 = (uint32_t)acc[3];
        const uint32_t hi = __builtin_amdgcn_perm(h1, h0, 0x0c0c0501u) | __builtin_amdgcn_perm(h3, h2, 0x05010c0cu);
        const uint32_t lo = (__builtin_amdgcn_perm(h1, h0, 0x0c0c0400u) | __builtin_amdgcn_perm(h3, h2, 0x04000c0cu)) ^ 0x80808080u;
        uint8_t* const wp = w_hi + 16 * (ROT ? (mt + rot) & 7 : mt);
        *reinterpret_cast<uint32_t*>(wp) = hi;
        *reinterpret_cast<uint32_t*>(wp + kPlane) = lo;
    }
    lds_barrier();

    stamp();
    // ---- 3. column pass: wave w owns column tile w; result -> staging tile (aliases the source tile) ----------
    if (IPX_PHASE_ON(g, 4))
#pragma unroll
    for (int nt = 0; nt < TY / 16; ++nt) {
        constexpr int c0 = 32768 * 256 + 32768 + 32768;          // rounding and the two +2^15 offsets, via the accumulator
        const v4i z = {0, 0, 0, 0}, zc = {c0, c0, c0, c0};
        const int vo = 16 * (ROT ? (nt + rrot) & 7 : nt + rrot);
        const v4i dhi = __builtin_amdgcn_mfma_i32_16x16x64_i8(lds16(a_vhi + vo), bv, z, 0, 0, 0);
        const v4i dlo = __builtin_amdgcn_mfma_i32_16x16x64_i8(lds16(a_vhi + kPlane + vo), bv, zc, 0, 0, 0);
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = ((uint32_t)dhi[i] << 8) + (uint32_t)dlo[i];
        // o[i] bits 16..23 = output byte of column 16 w + 4 gq + i, row 16 nt + m
        const uint32_t pk = __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0602u) | __builtin_amdgcn_perm(o[3], o[2], 0x06020c0cu);
        *reinterpret_cast<uint32_t*>(w_out + 16 * nt * kPitchO) = pk;
    }
    lds_barrier();

    stamp();
    // ---- 4. staging tile -> global, 16 bytes per lane, whole 128-byte rows -----------------------------------
    if (IPX_PHASE_ON(g, 8)) {
        const int x0 = tile_x(cur) * TX, y0 = tile_y(cur) * TY;
        uint8_t* dbase = dst + (int64_t)cur.img * g.d_img + (int64_t)y0 * g.d_row + x0;
        if (g.fast32 && y0 + TY <= g.h && x0 + TX <= g.L) {       // whole tile inside (block-uniform)
            st16_nt(dbase + ooff, *reinterpret_cast<const u32x4*>(r_out));
        } else {
            const int y = y0 + orow, x = x0 + 16 * occ;
            if (y < g.h && x < g.L) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(r_out);
                uint8_t* dp = dbase + (int64_t)orow * g.d_row + 16 * occ;
                if (x + 16 <= g.L) {
                    st16_nt(dp, v);
                } else {
                    for (int b = 0; x + b < g.L; ++b) dp[b] = (uint8_t)(v[b >> 2] >> (8 * (b & 3)));
                }
            }
        }
    }
    lds_barrier();                                              // the staging tile is the next source tile
    stamp();
    };
    while (pa.P < g.npatches) {
        const Pos fa = advance(pa);
        do_tile(pa, fa, bufa);
        pa = fa;
    }
}

template <int C>
void launch_c(hipStream_t st, const uint8_t* src, uint8_t* dst, const MGeom& g, const MTaps& t)
{
    const int hl = (g.r * C + 15) & ~15;
    const dim3 grid(8u * (unsigned)(g.pw * g.ph)), block(kThreads);
    const bool five = 64 + 2 * g.r <= 80;
#define IPX_MFMA_LAUNCH(HLV)                                                                                               \
    do {                                                                                                                   \
        if (five) hipLaunchKernelGGL((sep_mfma_kernel<C, HLV, 5>), grid, block, 0, st, src, dst, g, t);                     \
        else hipLaunchKernelGGL((sep_mfma_kernel<C, HLV, 6>), grid, block, 0, st, src, dst, g, t);                          \
    } while (0)
    switch (hl) {
        case 16: IPX_MFMA_LAUNCH(16); break;
        case 32: IPX_MFMA_LAUNCH(32); break;
        case 48: IPX_MFMA_LAUNCH(48); break;
        default: IPX_MFMA_LAUNCH(64); break;
    }
#undef IPX_MFMA_LAUNCH
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
    // ipx_sepwave.hip (one wave per column strip) is 1.4-3.5x faster from 1080p frames up; inputs of a few hundred rows and
    // columns, where its frame pieces are most of the work, stay with the tile kernel below (scripts/dev/wave_shapes.py)
    bool wave_form = (int64_t)n * h * w * c >= (2 << 20);
#ifdef IPX_DEV
    if (const char* e = getenv("IPX_WAVE")) wave_form = atoi(e) != 0;
#endif
    if (wave_form) {
        const int rc = ipx_sep_wave(st, src, dst, n, h, w, c, s_row, s_img, d_row, d_img, kx, ky, k, top, bot);
        if (rc <= 0) return rc;
    }
    MGeom g;
    g.h = h; g.w = w; g.L = w * c; g.r = k / 2; g.K = k;
    g.s_row = s_row; g.s_img = s_img; g.d_row = d_row; g.d_img = d_img; g.top = top; g.bot = bot;
    g.fast32 = s_row < (1 << 24) && d_row < (1 << 24) && s_row >= 0 && d_row >= 0;
    g.tiles_x = (g.L + TX - 1) / TX;
    g.tiles_y = (h + TY - 1) / TY;
    // blocks per XCD: 96 (3 resident blocks of 52-55 KiB LDS on each of its 32 CUs), fewer for small inputs; patch
    // shape = the divisor pair of that number that wastes the fewest tile slots, squarest first
    const uint64_t nt = (uint64_t)n * g.tiles_x * g.tiles_y;
    int nb = 96;
#ifdef IPX_DEV
    if (const char* e = getenv("IPX_MFMA_BLOCKS")) { if (atoi(e) > 0) nb = atoi(e); }
#endif
    if ((uint64_t)nb * 8 > nt) nb = (int)((nt + 7) / 8);
    g.pw = nb < g.tiles_x ? nb : g.tiles_x;
    g.ph = nb / g.pw > 0 ? nb / g.pw : 1;
    if (nb == 96) {
        static const int cand[][2] = {{12, 8}, {8, 12}, {16, 6}, {6, 16}, {24, 4}, {4, 24}, {32, 3}, {48, 2}, {96, 1}};
        uint64_t best = ~0ull;
        for (const auto& cd : cand) {
            const uint64_t padded = (uint64_t)((g.tiles_x + cd[0] - 1) / cd[0] * cd[0]) * ((g.tiles_y + cd[1] - 1) / cd[1] * cd[1]);
            if (padded < best) { best = padded; g.pw = cd[0]; g.ph = cd[1]; }
        }
    }
    g.patches_x = (g.tiles_x + g.pw - 1) / g.pw;
    g.patches_y = (g.tiles_y + g.ph - 1) / g.ph;
    const uint64_t np = (uint64_t)n * g.patches_x * g.patches_y;
    if (np > 0x7ffffff0ull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: image too large", fn);
    g.npatches = (unsigned)np;
    MTaps t{};
#ifdef IPX_DEV
    g.prof = nullptr;
    g.ablate = getenv("IPX_MFMA_ABLATE") ? atoi(getenv("IPX_MFMA_ABLATE")) : 0;
    const bool prof = getenv("IPX_MFMA_PROF") != nullptr;                    // print block 0's phase stamps
    if (prof) { IPX_HIP(hipMalloc((void**)&g.prof, 64 * sizeof(long long))); IPX_HIP(hipMemset(g.prof, 0, 64 * sizeof(long long))); }
#endif
    for (int i = 0; i < k; ++i) { t.kx[i] = (uint8_t)kx[i]; t.ky[i] = (uint8_t)ky[i]; }
    switch (c) {
        case 1: launch_c<1>(st, src, dst, g, t); break;
        case 2: launch_c<2>(st, src, dst, g, t); break;
        case 3: launch_c<3>(st, src, dst, g, t); break;
        default: launch_c<4>(st, src, dst, g, t); break;
    }
    IPX_LAUNCH_CHECK("sep_mfma_kernel");
#ifdef IPX_DEV
    if (prof) {
        long long hst[64];
        IPX_HIP(hipDeviceSynchronize());
        IPX_HIP(hipMemcpy(hst, g.prof, sizeof(hst), hipMemcpyDeviceToHost));
        IPX_HIP(hipFree(g.prof));
        fprintf(stderr, "sep_mfma k=%d stamps (cycles since first):", k);
        for (int i = 1; i < 64 && hst[i]; ++i) fprintf(stderr, " %lld", hst[i] - hst[0]);
        fprintf(stderr, "\n");
    }
#endif
    return IPX_OK;
}

// ipx_sepwave.hip -- separable fixed-point filter for 9..31 taps on the matrix cores, one WAVE per column strip
// (cv2.GaussianBlur with a large kernel, server.py:428; BASELINE.json configs[3]: 15x15 and 31x31 on 8K RGB).
//
// Arithmetic as in ipx_sepmfma.hip (both passes are banded-Toeplitz products on v_mfma_i32_16x16x64_i8, exact in int32, source
// bytes as u - 128, Q8.8 row results split into a high and a low byte plane).  What differs is who owns what:
//
//   wave   = a strip of 128 byte columns (8 column tiles of 16; interleaved channels stay interleaved, taps sit C bytes
//            apart) walking DOWN a segment of rows, 16 source rows per step.  Nothing is shared between waves: no block
//            barrier after the prologue, so the waves of a CU drift apart and their LDS, matrix and vector work overlap.
//   stage  : the step's 16 x (128 + 2 HL) source bytes are loaded into registers ONE STEP AHEAD (plain 16-byte loads, the
//            row offsets of a lane never change: a wave-uniform base moves down), flipped to signed (u ^ 0x80) and written
//            to a wave-private two-slot LDS ring.
//   row    : A = ds_read_b128 of the source rows (neighbouring column tiles share 64-byte fragments: 12 reads feed 16
//            products), B = Toeplitz(kx) in registers.  The accumulator of a 16 x 16 tile, lane (column, gq) holding rows
//            4 gq .. 4 gq + 3, IS the column pass's A fragment once the 64 k-slots of that product are ordered to match
//            (slot (gq, 4 t + i) = row 4 gq + i of source tile t of a window of four): the high and low bytes of the last
//            four source tiles stay in two 4-dword register windows per column tile, nothing is transposed through LDS.
//   col    : two products per column tile (high plane, low plane) with Toeplitz(ky) in that k-slot order,
//            out = (256 Dhi + Dlo + c0) >> 16 (the offsets of the u - 128 trick folded into c0; taps sum to 256).
//   store  : a lane ends up with 4 output bytes per column tile; a 4 x 4 dword transpose across the lane quarters
//            (v_permlane32_swap + v_permlane16_swap) turns four tiles into 16 contiguous bytes per lane: two 16-byte
//            stores per step write 16 rows x 128 bytes as 64-byte row pieces.
// Work items.  The INTERIOR of an image (strips whose source window lies inside the row, 16-row tiles whose source rows lie
// inside the image) is cut into long segments that run a branch-free loop (4 steps per iteration so that the window
// slots are static registers).  The FRAME around it -- the first / last strips, the top tile row, the last one or two tile
// rows: REFLECT_101, the caller's halo rows, ragged widths and heights -- is cut into short pieces that take a generic
// loop (same arithmetic; per-step addresses, the out-of-row bytes of a strip filled in from the strip's own LDS rows,
// predicated stores).  The frame is 1-3 % of a large image; its blocks come last in the grid (short pieces that fill the
// wave slots the interior blocks leave).
#include "ipx_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));

constexpr int NT = 8, TXW = 16 * NT;                 // column tiles / output byte columns per wave
constexpr int kWaves = 4, kThreads = 64 * kWaves;
constexpr int kFrameTiles = 4;                       // 16-row tiles per frame piece

struct WGeom {
    int h, w, L, r, K, C;
    int64_t s_row, s_img, d_row, d_img;
    int nstrips, ntiles;                             // strips of 128 bytes, 16-row tiles per image
    int sl, sr, ta, tb;                              // interior: strips [sl, sr) x tiles [ta, tb)
    int seg_tiles;                                   // interior segments: tiles per segment
    int nbp, nsp;                                    // frame pieces per strip: below the interior / beside it
    unsigned g_items, f_items;                       // frame / interior wave items per image
    unsigned g_total, f_total;                       // ... over the batch
    unsigned g_blocks, f_blocks;                     // blocks of 4 wave items over the batch
    const uint8_t* top;
    const uint8_t* bot;
#ifdef IPX_DEV
    int dev_mode;                                    // dev build: 1 = skip the frame waves, 2 = skip the interior waves (timing only)
#endif
};
// kxe[d]: the row taps spread over the byte distance d = t * C (0 where d is not a multiple of C); ky as given
struct WTaps { uint8_t kxe[128]; uint8_t ky[32]; };

__device__ __forceinline__ v4i lds16(const uint8_t* p) { return *reinterpret_cast<const v4i*>(p); }
__device__ __forceinline__ void wave_lds_sync()      // orders this wave's LDS writes before its other lanes' reads
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int V> using ic = std::integral_constant<int, V>;

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Vector-memory operations of a wave that are younger than the DMAs of source tile S + 1 at the top of step S (see the
// interior loop): the DMAs of tiles S + 2 .. S + RING - 1 and two stores for every step s >= LAG in [S + 1 - RING, S - 1].
template <int RING, int NLD, int LAG> constexpr int wait_n(int S)
{
    int lo = S + 1 - RING;
    if (lo < LAG) lo = LAG;
    const int steps = S - lo > 0 ? S - lo : 0;
    return (RING - 2) * NLD + 2 * steps;
}

template <int HL, int LAG>
__global__ __launch_bounds__(kThreads, 2) void sep_wave_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                               WGeom g, WTaps taps)
{
    constexpr int KSH = (16 + 2 * HL + 63) / 64;                 // k-steps of the row pass
    constexpr int SW = TXW + 2 * HL, NCH = SW / 16;              // source bytes / 16-byte chunks per strip row
    constexpr int PS = SW % 64 == 32 ? SW : SW + 32;             // LDS pitch = 32 mod 64: conflict-free ds_read_b128 fragments
    constexpr int NLD = (16 * NCH + 63) / 64;                    // 16-byte loads per lane and step
    constexpr int OFF = LAG == 2 ? 16 : 8;                       // source tile 0 starts OFF rows above the segment
    constexpr int kSlot = 16 * PS;
    constexpr int RING = HL <= 48 ? 3 : 2;                       // LDS-DMA ring of the interior loop: RING - 1 source tiles in flight
    constexpr int kRaw = 1024 * NLD;                             // a source tile as it arrives (lane order: chunk q at byte 16 q)
    // 2 slots | 64: the last fragment of a row reads 16 bytes past it (zero taps) | the ring (frame pieces: their mirror table)
    constexpr int kWaveLds = 2 * kSlot + 64 + RING * kRaw;
    // row taps: d0 >= -HL - 15, d0 + 15 <= 64 KSH + 47 - HL + r C; column taps: i0 >= -55, i0 + 3 <= 63 + 15
    constexpr int kPadX = 96, kPadX2 = 192, kPadY = 64, kTapsLds = kPadX + 128 + kPadX2 + kPadY + 32 + 64;
    __shared__ __attribute__((aligned(16))) uint8_t s_all[kWaves * kWaveLds + kTapsLds];
    uint8_t* const s_taps = s_all + kWaves * kWaveLds;           // 0 .. | kxe[128] | 0 .. | ky[32] | 0 ..

    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, gq = lane >> 4;
    const int r = g.r, C = g.C;

    // taps -> LDS, zero-padded on both sides, so that a fragment of a Toeplitz operand is a run of table bytes
    for (int i = tid; i < kTapsLds; i += kThreads) {
        const int dx = i - kPadX, dy = i - (kPadX + 128 + kPadX2) - kPadY;
        s_taps[i] = (unsigned)dx < 128u ? taps.kxe[dx] : (unsigned)dy < 32u ? taps.ky[dy] : (uint8_t)0;
    }
    __syncthreads();

    // ---- Toeplitz operands (registers, the same for every step) --------------------------------------------------
    v4i bh[KSH], bv[4];
    {
        const int n = m;                                          // B column = lane & 15
        const uint8_t* const tx = s_taps + kPadX;                 // tx[d]: row tap at byte distance d from the first tap
        const uint8_t* const ty = s_taps + kPadX + 128 + kPadX2 + kPadY;   // ty[i]: column tap i
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
            // k-slot (gq, jj) is source byte 64 ks + 16 gq + jj of the tile's window: distance d0 + jj from the first tap
            const int d0 = 64 * ks + 16 * gq - HL - n + r * C;
            uint32_t wds[4];
            __builtin_memcpy(wds, tx + d0, 16);
            bh[ks] = (v4i){(int)wds[0], (int)wds[1], (int)wds[2], (int)wds[3]};
        }
        // column pass: source tile S lives in dword S & 3 of the window; at step S (tt = S & 3) dword d holds window
        // position (d - tt - 1) & 3 (0 = oldest); tap i of output row n sits at window row OFF + 16 (3 - LAG) + n - r + i
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            uint32_t wds[4];
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
                const int wp = (d4 - tt - 1) & 3;
                const int i0 = 16 * wp + 4 * gq - (OFF + 16 * (3 - LAG)) - n + r;   // taps i0 .. i0 + 3
                __builtin_memcpy(&wds[d4], ty + i0, 4);
            }
            bv[tt] = (v4i){(int)wds[0], (int)wds[1], (int)wds[2], (int)wds[3]};
        }
    }

    // ---- this wave's item: image, strip, first tile, tiles -------------------------------------------------------
    bool fast;
    int strip, t0, nT;
    unsigned img;
    if (blockIdx.x >= g.f_blocks) {                               // frame pieces (after the interior blocks), dealt round-robin over the XCDs
        fast = false;
        const unsigned wi = (blockIdx.x - g.f_blocks) * kWaves + wave;
        if (wi >= g.g_total) return;                              // (after the only block barrier)
        img = wi / g.g_items;
        unsigned it = wi - img * g.g_items;
        const unsigned n_top = (unsigned)(g.nstrips * g.ta), n_bot = (unsigned)(g.nstrips * g.nbp);
        if (it < n_top) {                                         // the tile row above the interior
            strip = (int)it; t0 = 0; nT = g.ta;
        } else if (it < n_top + n_bot) {                          // the tile rows below it
            it -= n_top;
            strip = (int)(it % (unsigned)g.nstrips);
            t0 = g.tb + kFrameTiles * (int)(it / (unsigned)g.nstrips);
            nT = g.ntiles - t0 < kFrameTiles ? g.ntiles - t0 : kFrameTiles;
        } else {                                                  // the strips beside it
            it -= n_top + n_bot;
            const int ss = (int)(it / (unsigned)g.nsp), piece = (int)(it - (unsigned)ss * (unsigned)g.nsp);
            strip = ss < g.sl ? ss : g.sr + (ss - g.sl);
            t0 = g.ta + kFrameTiles * piece;
            nT = g.tb - t0 < kFrameTiles ? g.tb - t0 : kFrameTiles;
        }
    } else {                                                      // interior segments: neighbouring strips share an XCD's L2
        fast = true;
        const unsigned wi = xcd_remap(blockIdx.x, g.f_blocks) * kWaves + wave;
        if (wi >= g.f_total) return;
        img = wi / g.f_items;
        const unsigned it = wi - img * g.f_items;
        const int ns = g.sr - g.sl;
        const int seg = (int)(it / (unsigned)ns);
        strip = g.sl + (int)(it - (unsigned)seg * (unsigned)ns);
        t0 = g.ta + seg * g.seg_tiles;
        nT = g.tb - t0 < g.seg_tiles ? g.tb - t0 : g.seg_tiles;
    }
    strip = __builtin_amdgcn_readfirstlane(strip);
    t0 = __builtin_amdgcn_readfirstlane(t0);
    nT = __builtin_amdgcn_readfirstlane(nT);
    img = (unsigned)__builtin_amdgcn_readfirstlane((int)img);
    const int nS = nT + LAG;                                      // source tiles (steps)
    if (nS < 4) fast = false;                                     // the branch-free loop starts with four steps
#ifdef IPX_DEV
    if (((g.dev_mode & 3) == 1 && !fast) || ((g.dev_mode & 3) == 2 && fast) || (g.dev_mode & 3) == 3) return;
#endif
    const int x0 = strip * TXW;
    const int y0 = 16 * t0, y1 = y0 + 16 * nT < g.h ? y0 + 16 * nT : g.h;
    const int Y0 = y0 - OFF, xs0 = x0 - HL;
    const uint8_t* const simg = src + (int64_t)img * g.s_img;
    uint8_t* const dimg = dst + (int64_t)img * g.d_img;

    uint8_t* const wlds = s_all + wave * kWaveLds;
    uint8_t* const wraw = wlds + 2 * kSlot + 64;
    const uint8_t* const a_base = wlds + m * PS + 16 * gq;        // row pass A: + slot, + 16 f for fragment f
    uint32_t whi[NT][4], wlo[NT][4];                              // register windows: high / low bytes of the last 4 source tiles
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) { whi[j][t] = 0; wlo[j][t] = 0; }

    // One step: row pass of the source tile in LDS slot `slot`, its bytes into window dword TT, column pass of the output
    // tile LAG tiles back, transposition; `store(hf, v)` receives this lane's 16 bytes of the output tile:
    // row 8 hf + (lane & 7), byte column 64 ((lane >> 3) & 1) + 16 (lane >> 4) of the strip.
    // The eight column tiles go through a software pipeline, one tile per time slot:
    //   slot t:  F(t + 2) fragment reads | R(t) row products | P(t - 1) pack into the window, C(t - 1) column products |
    //            O(t - 2) output bytes (+ transposition and store after tiles 3 and 7)
    // so that every consumer finds its producer one slot back (LDS and matrix-core latencies are covered by the other
    // stages' instructions instead of being waited out).  NEXTF: the fragments of tiles 0 and 1 of the NEXT step are
    // read in the last two slots (the interior loop has that tile staged by then); `mid()` runs in slot 3.
    v4i frc[2][KSH];                                              // fragments carried into the next step (NEXTF)
    auto step = [&](auto tt_c, auto col_c, auto nextf_c, int slot, auto&& store, auto&& mid) {
        constexpr int TT = decltype(tt_c)::value;
        constexpr bool COL = decltype(col_c)::value != 0, NEXTF = decltype(nextf_c)::value != 0;
        const uint8_t* const ab = a_base + slot * kSlot;
        const uint8_t* const abn = a_base + (slot ^ 1) * kSlot;
        v4i fr[NT][KSH], acc[NT], dhi[NT], dlo[NT];
        uint32_t od[NT];
        u32x4 ot[NT / 4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks) fr[j][ks] = NEXTF ? frc[j][ks] : lds16(ab + 16 * (j + 4 * ks));
#pragma unroll
        for (int t = 0; t < NT + 2; ++t) {
            if (t + 2 < NT) {
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks) fr[t + 2][ks] = lds16(ab + 16 * (t + 2 + 4 * ks));
            } else if (NEXTF) {
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks) frc[t + 2 - NT][ks] = lds16(abn + 16 * (t + 2 - NT + 4 * ks));
            }
            if (t < NT) {
                v4i a = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks) a = __builtin_amdgcn_mfma_i32_16x16x64_i8(fr[t][ks], bh[ks], a, 0, 0, 0);
                acc[t] = a;
            }
            if (t >= 1 && t - 1 < NT) {
                const int j = t - 1;
                // acc[i] = row 4 gq + i, column m; Q8.8 value h = acc + 2^15: (h >> 8) - 128 is byte 1 of acc, (h & 255) - 128 = byte 0 ^ 0x80
                const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)acc[j][1], (uint32_t)acc[j][0], 0x05040100u);
                const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)acc[j][3], (uint32_t)acc[j][2], 0x05040100u);
                whi[j][TT] = __builtin_amdgcn_perm(p23, p01, 0x07050301u);
                wlo[j][TT] = __builtin_amdgcn_perm(p23, p01, 0x06040200u) ^ 0x80808080u;
                if (COL) {
                    constexpr int c0 = 32768 * 256 + 32768 + 32768;   // rounding and the two +2^15 offsets, via the accumulator
                    const v4i z = {0, 0, 0, 0}, zc = {c0, c0, c0, c0};
                    const v4i ahi = {(int)whi[j][0], (int)whi[j][1], (int)whi[j][2], (int)whi[j][3]};
                    const v4i alo = {(int)wlo[j][0], (int)wlo[j][1], (int)wlo[j][2], (int)wlo[j][3]};
                    dhi[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ahi, bv[TT], z, 0, 0, 0);
                    dlo[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(alo, bv[TT], zc, 0, 0, 0);
                }
            }
            if (COL && t >= 2) {
                const int j = t - 2;
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = ((uint32_t)dhi[j][i] << 8) + (uint32_t)dlo[j][i];
                // o[i] bits 16..23 = output byte of column 16 j + 4 gq + i, row m of the output tile
                od[j] = __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0602u) | __builtin_amdgcn_perm(o[3], o[2], 0x06020c0cu);
                if ((j & 3) == 3) {
                    const int J = j >> 2;
                    // X[t][q] (register t = tile 4 J + t, lane quarter q = columns 4 q .. 4 q + 3) -> X[q][t]
                    u32x2v a = __builtin_amdgcn_permlane32_swap(od[4 * J + 0], od[4 * J + 2], false, false);
                    u32x2v b = __builtin_amdgcn_permlane32_swap(od[4 * J + 1], od[4 * J + 3], false, false);
                    u32x2v c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
                    u32x2v d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
                    ot[J] = (u32x4){c[0], c[1], d[0], d[1]};      // bytes 64 J + 16 (lane >> 4) .. + 15 of row lane & 15
                }
                if (j == NT - 1) {
                    // rows n and n + 8 trade halves (lanes i and i ^ 8 of every 16): afterwards a store instruction covers 8 whole
                    // 128-byte rows instead of 16 half rows.  A: rows 0..7, lane (n, q) -> row n & 7, bytes 64 (n >> 3) + 16 q ..;
                    // B: the same lanes, rows 8..15.
                    u32x4 va, vb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        va[e] = (uint32_t)__builtin_amdgcn_update_dpp((int)ot[0][e], (int)ot[1][e], 0x128, 0xf, 0xc, false);   // row_ror:8 into lanes 8..15
                        vb[e] = (uint32_t)__builtin_amdgcn_update_dpp((int)ot[1][e], (int)ot[0][e], 0x128, 0xf, 0x3, false);   // ... into lanes 0..7
                    }
                    store(0, va);
                    store(1, vb);
                }
            }
            if (t == 3) mid();
#ifndef IPX_WAVE_NOSB
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    auto nomid = [] {};

    if (__builtin_amdgcn_readfirstlane((int)fast)) {
        // ---- interior: source rows by LDS-DMA, RING - 1 steps ahead, hand-counted waits, no branch in the loop -----------
        // A step's 16 x NCH chunks arrive in lane order (chunk q = 64 i + lane at byte 16 q of a ring slot: LDS-DMA writes
        // base + 16 lane); a copy pass, one step ahead, flips them to signed and lays them out at the fragment pitch.
        // No lane predicates here: a DMA or copy inside a divergent branch gets duplicated per branch arm by the compiler, which
        // breaks the instruction counts the waits below rely on.  The lanes past a step's last chunk (half a wave when 16 NCH is
        // not a multiple of 64) re-load the last chunk into the padding of the ring slot and copy it to the slack bytes.
        unsigned goff[NLD], loff[NLD];
        int crow[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = 64 * i + lane;
            const int qc = q < 16 * NCH ? q : 16 * NCH - 1;
            const int row = qc / NCH, ch = qc - row * NCH;
            crow[i] = row;
            goff[i] = (unsigned)(row * (int)g.s_row + 16 * ch);
            loff[i] = q < 16 * NCH ? (unsigned)(row * PS + 16 * ch) : (unsigned)(2 * kSlot);
        }
        const int64_t dstep = 16 * g.d_row;
#ifdef IPX_DEV
        const bool dev_nost = (g.dev_mode & 4) != 0, dev_nold = (g.dev_mode & 8) != 0;   // timing only: results wrong
#endif
        auto dma = [&](int s, int rs) {                           // source tile s -> ring slot rs (= s mod RING, wave-uniform)
            uint8_t* const slot = wraw + rs * kRaw;
#ifdef IPX_DEV
            if (dev_nold) return;
#endif
            const int ytop = Y0 + 16 * (s < nS ? s : nS - 1);     // (past the last tile: re-load it, nobody reads that slot)
            if (ytop >= 0 && ytop + 16 <= g.h) {                  // wave-uniform: both arms issue NLD DMAs
                const uint8_t* b = simg + (int64_t)ytop * g.s_row + xs0;
#pragma unroll
                for (int i = 0; i < NLD; ++i)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(b + goff[i]),
                                                     (void __attribute__((address_space(3)))*)(slot + 1024 * i), 16, 0, 0);
            } else {                                              // a tile across the top / bottom edge: REFLECT_101 or the caller's halo rows
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    int ys = ytop + crow[i];
                    ys = ys < -r ? -r : ys > g.h - 1 + r ? g.h - 1 + r : ys;      // farther out: only meets zero taps
                    const uint8_t* rp = halo_row(simg, g.top, g.bot, ys, g.h, r, g.s_row);
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(rp + xs0 + (goff[i] - (unsigned)(crow[i] * (int)g.s_row))),
                                                     (void __attribute__((address_space(3)))*)(slot + 1024 * i), 16, 0, 0);
                }
            }
        };
        u32x4 cv[NLD];
        auto copy_read = [&](int rs) {
            const uint8_t* const slot = wraw + rs * kRaw + 16 * lane;
#pragma unroll
            for (int i = 0; i < NLD; ++i) cv[i] = *reinterpret_cast<const u32x4*>(slot + 1024 * i);
        };
        auto copy_write = [&](int slot) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4*>(wlds + (loff[i] < 2u * kSlot ? slot * kSlot : 0) + loff[i]) = cv[i] ^ (u32x4)(0x80808080u);
        };
        const unsigned ooff = (unsigned)((m & 7) * (int)g.d_row + 64 * (m >> 3) + 16 * gq);
        const int64_t dhalf = 8 * g.d_row;
        uint8_t* dT = dimg + (int64_t)(y0 - 16 * LAG) * g.d_row + x0;   // output tile of step S: rows y0 + 16 (S - LAG) ..
#ifdef IPX_DEV
        auto st = [&](int hf, u32x4 v) { if (!dev_nost) st16_nt(dT + (hf ? dhalf : 0) + ooff, v); };
#else
        auto st = [&](int hf, u32x4 v) { st16_nt(dT + (hf ? dhalf : 0) + ooff, v); };
#endif
        auto none = [](int, u32x4) {};
        // This wave's vector-memory queue holds its DMAs (NLD per source tile) and its stores (2 per step from step LAG on),
        // in issue order.  At the top of step S the tile S + 1 must have landed; younger than its DMAs are those of tiles
        // S + 2 .. S + RING - 1 and the stores of the steps since it was issued (step S + 1 - RING): vmcnt(N(S)).
        int r0 = 0, r1 = 1 % RING;                                // ring slots of tiles S and S + 1
#pragma unroll
        for (int s = 0; s < RING; ++s) dma(s, s);
        wait_vmcnt<(RING - 1) * NLD>();
        copy_read(0);
        copy_write(0);
        auto head = [&](auto n_c, int S) {                        // top of step S
            wait_vmcnt<decltype(n_c)::value>();
            copy_read(r1);
            dma(S + RING, r0);
            r0 = r1; r1 = r1 + 1 == RING ? 0 : r1 + 1;
        };
        constexpr int kSteady = (RING - 2) * NLD + 2 * (RING - 1);
        // first four steps: the window fills, tiles above the segment are not computed
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks) frc[j][ks] = lds16(a_base + 16 * (j + 4 * ks));
        auto cw0 = [&] { copy_write(0); };
        auto cw1 = [&] { copy_write(1); };
        head(ic<wait_n<RING, NLD, LAG>(0)>{}, 0); step(ic<0>{}, ic<0>{}, ic<1>{}, 0, none, cw1); dT += dstep;
        head(ic<wait_n<RING, NLD, LAG>(1)>{}, 1);
        if constexpr (LAG <= 1) step(ic<1>{}, ic<1>{}, ic<1>{}, 1, st, cw0); else step(ic<1>{}, ic<0>{}, ic<1>{}, 1, none, cw0);
        dT += dstep;
        head(ic<wait_n<RING, NLD, LAG>(2)>{}, 2); step(ic<2>{}, ic<1>{}, ic<1>{}, 0, st, cw1); dT += dstep;
        head(ic<wait_n<RING, NLD, LAG>(3)>{}, 3); step(ic<3>{}, ic<1>{}, ic<1>{}, 1, st, cw0); dT += dstep;
        static_assert(wait_n<RING, NLD, LAG>(4) == kSteady, "the loop's wait count");
        int S = 4;
        for (; S + 4 <= nS; S += 4) {
            head(ic<kSteady>{}, S); step(ic<0>{}, ic<1>{}, ic<1>{}, 0, st, cw1); dT += dstep;
            head(ic<kSteady>{}, S + 1); step(ic<1>{}, ic<1>{}, ic<1>{}, 1, st, cw0); dT += dstep;
            head(ic<kSteady>{}, S + 2); step(ic<2>{}, ic<1>{}, ic<1>{}, 0, st, cw1); dT += dstep;
            head(ic<kSteady>{}, S + 3); step(ic<3>{}, ic<1>{}, ic<1>{}, 1, st, cw0); dT += dstep;
        }
        const int rem = nS - S;                                   // 0..3 steps left (wave-uniform)
        if (rem > 0) { head(ic<kSteady>{}, S); step(ic<0>{}, ic<1>{}, ic<1>{}, 0, st, cw1); dT += dstep; }
        if (rem > 1) { head(ic<kSteady>{}, S + 1); step(ic<1>{}, ic<1>{}, ic<1>{}, 1, st, cw0); dT += dstep; }
        if (rem > 2) { head(ic<kSteady>{}, S + 2); step(ic<2>{}, ic<1>{}, ic<0>{}, 0, st, nomid); }
        wait_vmcnt<0>();                                          // the DMAs issued past the last tile land before LDS is released
        return;
    }

    // ---- frame pieces: image edges, halo rows, ragged strips and rows ------------------------------------------------
    const bool x_in = xs0 >= 0 && xs0 + SW <= g.L;                 // wave-uniform
    uint16_t* const fixt = reinterpret_cast<uint16_t*>(wraw);    // strip byte -> the strip byte it mirrors (2 SW <= 512 bytes)
    const int nl = xs0 < 0 ? (-xs0 < SW ? -xs0 : SW) : 0;         // strip bytes [0, nl) lie left of the row
    const int nr = g.L - xs0 < SW ? (g.L - xs0 > 0 ? g.L - xs0 : 0) : SW;   // strip bytes [nr, SW) lie right of it
    if (!x_in) {
        // bytes outside the row: REFLECT_101 in pixel space, later copied from the strip's own LDS rows (the source of every
        // byte an output of this strip needs lies within HL bytes of that output, i.e. inside the window; others stay as they are)
        for (int b = lane; b < SW; b += 64) {
            const int col = xs0 + b;
            int sb = 0xffff;
            if (col < 0 || col >= g.L) {
                int px = col >= 0 ? col / C : -((-col + C - 1) / C);
                const int chn = col - px * C;
                px = px < -r ? -r : px > g.w - 1 + r ? g.w - 1 + r : px;       // farther out: never used
                sb = reflect101(px, g.w) * C + chn - xs0;
                if (sb < 0 || sb >= SW) sb = 0xffff;
            }
            fixt[b] = (uint16_t)sb;
        }
        wave_lds_sync();
    }
    const int sh = (16 - (g.L & 15)) & 15;                        // the chunk a row ends in is loaded sh bytes early
    u32x4 R[NLD];
    int ln = lane;                                                // re-made opaque every step: what derives from it is recomputed, not kept live
    auto issue_g = [&](int S) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int q = 64 * i + ln;
            q = q < 16 * NCH ? q : 16 * NCH - 1;
            const int row = q / NCH, ch = q - row * NCH;
            int ys = Y0 + 16 * S + row;
            ys = ys < -r ? -r : ys > g.h - 1 + r ? g.h - 1 + r : ys;          // farther out: only meets zero taps / unused rows
            const uint8_t* rp = halo_row(simg, g.top, g.bot, ys, g.h, r, g.s_row);
            int gx = xs0 + 16 * ch;
            gx = gx < 0 ? 0 : gx > g.L - 16 ? g.L - 16 : gx;                  // (L >= 16: checked on the host)
            R[i] = ld16(rp + gx);
        }
    };
    auto consume_g = [&](int slot) {
        uint8_t* const sp = wlds + slot * kSlot;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int q = 64 * i + ln;
            q = q < 16 * NCH ? q : 16 * NCH - 1;
            const int row = q / NCH, ch = q - row * NCH;
            const int gx = xs0 + 16 * ch;
            u32x4 v = R[i];
            if (sh && gx < g.L && gx + 16 > g.L) {                            // loaded from L - 16: move its bytes down by sh
                const uint32_t w[8] = {v[0], v[1], v[2], v[3], 0u, 0u, 0u, 0u};
                const int qw = sh >> 2;
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t lo = 0, hi = 0;
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2) if (qw == c2) { lo = w[e + c2]; hi = w[e + c2 + 1]; }
                    o[e] = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)sh);
                }
                v = o;
            }
            *reinterpret_cast<u32x4*>(sp + row * PS + 16 * ch) = v ^ (u32x4)(0x80808080u);
        }
        if (!x_in) {
            wave_lds_sync();
            uint8_t* const rowp = sp + m * PS;
            for (int b = gq; b < nl; b += 4) { const int sb = fixt[b]; if (sb != 0xffff) rowp[b] = rowp[sb]; }
            for (int b = nr + gq; b < SW; b += 4) { const int sb = fixt[b]; if (sb != 0xffff) rowp[b] = rowp[sb]; }
            wave_lds_sync();
        }
    };
    int Sg = 0;                                                   // the step being computed (for the store predicate)
    auto st_g = [&](int hf, u32x4 v) {
        const int y = y0 + 16 * (Sg - LAG) + 8 * hf + (m & 7), x = x0 + 64 * (m >> 3) + 16 * gq;
        if (y < y1 && x < g.L) {
            uint8_t* dp = dimg + (int64_t)y * g.d_row + x;
            if (x + 16 <= g.L) st16_nt(dp, v);
            else for (int b = 0; x + b < g.L; ++b) dp[b] = (uint8_t)(v[b >> 2] >> (8 * (b & 3)));
        }
    };
    auto none_g = [](int, u32x4) {};
    issue_g(0);
    for (Sg = 0; Sg < nS; ++Sg) {
        const int slot = Sg & 1;
        asm volatile("" : "+v"(ln));
        consume_g(slot);
        if (Sg + 1 < nS) issue_g(Sg + 1);
        if (Sg < LAG) {                                           // the window fills
            if ((Sg & 3) == 0) step(ic<0>{}, ic<0>{}, ic<0>{}, slot, none_g, nomid);
            else step(ic<1>{}, ic<0>{}, ic<0>{}, slot, none_g, nomid);
        } else {
            switch (Sg & 3) {
                case 0: step(ic<0>{}, ic<1>{}, ic<0>{}, slot, st_g, nomid); break;
                case 1: step(ic<1>{}, ic<1>{}, ic<0>{}, slot, st_g, nomid); break;
                case 2: step(ic<2>{}, ic<1>{}, ic<0>{}, slot, st_g, nomid); break;
                default: step(ic<3>{}, ic<1>{}, ic<0>{}, slot, st_g, nomid); break;
            }
        }
    }
}

template <int LAG>
void launch_hl(hipStream_t st, const uint8_t* src, uint8_t* dst, const WGeom& g, const WTaps& t, int hl)
{
    const dim3 grid(g.g_blocks + g.f_blocks), block(kThreads);
    unsigned dyn = 0;
#ifdef IPX_DEV
    if (const char* e = getenv("IPX_WAVE_DYNLDS")) dyn = (unsigned)atoi(e);   // dev: extra LDS per block, to lower the occupancy
#endif
    switch (hl) {
        case 16: hipLaunchKernelGGL((sep_wave_kernel<16, LAG>), grid, block, dyn, st, src, dst, g, t); break;
        case 32: hipLaunchKernelGGL((sep_wave_kernel<32, LAG>), grid, block, dyn, st, src, dst, g, t); break;
        case 48: hipLaunchKernelGGL((sep_wave_kernel<48, LAG>), grid, block, dyn, st, src, dst, g, t); break;
        default: hipLaunchKernelGGL((sep_wave_kernel<64, LAG>), grid, block, dyn, st, src, dst, g, t); break;
    }
}

}  // namespace

// Called by ipx_sep_mfma for 9 <= k <= 31 with taps <= 127 (arguments validated by ipx_sep_filter_q8_u8).
// Returns 1 when the shape is not for this kernel (row strides beyond 32-bit in-step offsets), 0 after the launch, < 0 on error.
int ipx_sep_wave(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                 int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                 const uint16_t* kx, const uint16_t* ky, int k, const uint8_t* top, const uint8_t* bot)
{
    if (s_row < 0 || d_row < 0 || s_row >= (1 << 26) || d_row >= (1 << 26) || w * c < 16) return 1;
    WGeom g;
    g.h = h; g.w = w; g.L = w * c; g.r = k / 2; g.K = k; g.C = c;
    g.s_row = s_row; g.s_img = s_img; g.d_row = d_row; g.d_img = d_img; g.top = top; g.bot = bot;
    const int lag = g.r <= 8 ? 1 : 2;
    const int hl = (g.r * c + 15) & ~15;
    g.nstrips = (g.L + TXW - 1) / TXW;
    g.ntiles = (h + 15) / 16;
    // interior: strips whose source window [x0 - hl, x0 + 128 + hl) lies inside the row, all their whole 16-row tiles (source
    // tiles across the top / bottom edge take the DMA's slow arm).  At least 4 tiles and one strip, else everything is frame.
    g.sl = (hl + TXW - 1) / TXW;
    g.sr = g.L >= hl + TXW ? (g.L - hl) / TXW : 0;
    g.ta = 0;
    g.tb = h / 16;
    if (g.sr <= g.sl || g.tb - g.ta < 4) { g.sl = g.sr = 0; g.ta = g.tb = 0; }
    const int ns = g.sr - g.sl, nt = g.tb - g.ta;
    // interior segments: aim at one round of resident blocks (2 blocks of 4 waves per CU = 2048 waves), at least 4 tiles each
    int tiles = 1;
    if (ns > 0) {
        const int64_t waves = 2048 - 2048 / 16;                   // (a little room for the frame pieces)
        int64_t segs = waves / ((int64_t)n * ns);
        if (segs < 1) segs = 1;
        tiles = (int)((nt + segs - 1) / segs);
        if (tiles < 4) tiles = 4;
#ifdef IPX_DEV
        if (const char* e = getenv("IPX_WAVE_TILES")) { if (atoi(e) > 0) tiles = atoi(e); }
#endif
        if (tiles > nt) tiles = nt;
    }
    g.seg_tiles = tiles;
    const int nfs = ns > 0 ? (nt + tiles - 1) / tiles : 0;
    g.nbp = (g.ntiles - g.tb + kFrameTiles - 1) / kFrameTiles;
    g.nsp = (nt + kFrameTiles - 1) / kFrameTiles;
    const uint64_t g_items = (uint64_t)g.nstrips * (g.ta + g.nbp) + (uint64_t)(g.nstrips - ns) * g.nsp;
    const uint64_t f_items = (uint64_t)ns * nfs;
    const uint64_t g_total = g_items * n, f_total = f_items * n;
    if (g_total + f_total > 0x7ffffff0ull) return ipx_fail(IPX_ERR_UNSUPPORTED, "ipx_sep_filter_q8_u8: image too large");
    g.g_items = (unsigned)g_items; g.f_items = (unsigned)f_items;
    g.g_total = (unsigned)g_total; g.f_total = (unsigned)f_total;
    g.g_blocks = (unsigned)((g_total + kWaves - 1) / kWaves);
    g.f_blocks = (unsigned)((f_total + kWaves - 1) / kWaves);
#ifdef IPX_DEV
    g.dev_mode = getenv("IPX_WAVE_MODE") ? atoi(getenv("IPX_WAVE_MODE")) : 0;
#endif
    WTaps t{};
    for (int i = 0; i < k; ++i) { t.kxe[i * c] = (uint8_t)kx[i]; t.ky[i] = (uint8_t)ky[i]; }   // (k - 1) c <= 120
    if (lag == 1) launch_hl<1>(st, src, dst, g, t, hl);
    else launch_hl<2>(st, src, dst, g, t, hl);
    IPX_LAUNCH_CHECK("sep_wave_kernel");
    return IPX_OK;
}

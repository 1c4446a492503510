// EXPERIMENT (round 2), not part of the library build.  Parity-green against the oracle (tests/test_gpu_parity.py with the
// dispatch in ipx_conv.hip pointed here), but not faster than ipx_sepmfma.hip on 2 x 8K BGR:
//   direct 4-byte stores from the accumulator layout   135 / 159 / 167 us (9 / 15 / 31 taps; old kernel 141 / 161 / 165),
//                                                      WRITE_SIZE 3.0x the output (16-byte pieces of 16 rows per store)
//   + LDS-staged 128-byte row stores (this file)       186 / 200 / 199 us, WRITE_SIZE 1.0x, but two barriers per phase and
//                                                      109 VGPRs + 52 KiB LDS = 2 blocks per CU
// What it established: (1) the k-slot ordering trick below works -- the row pass's accumulators ARE the column pass's A
// fragments, no transposition; (2) register prefetch behind any branch makes the compiler fall back to vmcnt(0) at
// every staging, LDS-DMA with hand-counted waits does not; (3) both MFMA kernels issue ~40 VALU instructions per
// 16 x 16 output tile and are bound by vector issue (SQ_WAIT_INST_ANY 63 %), not by the matrix cores (25 us of 160).
// Next step if resumed: one wave per 128-column strip (8 column tiles), wave-private LDS-DMA ring and output
// transposition, no block barrier at all.
// ipx_sepchain.hip -- separable fixed-point filter for 9..31 taps on the matrix cores, both passes chained through
// REGISTERS (cv2.GaussianBlur with a large kernel, server.py:428; BASELINE.json configs[3]: 15x15 and 31x31 on 8K RGB).
//
// Both passes of the filter are banded-Toeplitz matrix products (exact in int32 on v_mfma_i32_16x16x64_i8, see
// ipx_sepmfma.hip for the arithmetic).  What is new here is how the row pass hands its result to the column pass.  The
// row pass of a wave produces, per 16-row source tile, D[row 4 gq + i][column n] in lane (n, gq) -- and that IS the A
// fragment layout of the column pass, provided the 64 k-slots of the column-pass instruction are ORDERED to match:
// slot (gq, j) of lane (m = column, gq) is source row 16 (j / 4) + 4 gq + (j % 4) of a window of four consecutive
// 16-row tiles.  Sums are order-free in integers, so only the Toeplitz operand of the column pass has to be built in the
// same order (once per kernel, in registers).  The high and low bytes of the Q8.8 row results of the last four tiles
// therefore stay in two 4-dword register windows per lane; nothing is transposed through LDS and the only barrier left
// is the one behind the staging of the source rows.
//
//   block  = 8 waves = a strip of 128 byte columns (wave w owns columns 16 w .. 16 w + 15; interleaved channels stay
//            interleaved, taps sit C bytes apart), walking DOWN a segment of rows, 64 source rows per phase
//   phase  : the 64 x (128 + 2 HL) source bytes (prefetched into registers during the previous phase, REFLECT_101 /
//            halo rows applied at the image edges) -> LDS as u - 128; one LDS-only barrier; then per 16-row tile:
//            row pass (KSH MFMAs, A = ds_read_b128 of the source rows), bytes of the result into the register window,
//            column pass for the output tile two tiles back (2 MFMAs: high and low plane), 4 output bytes per lane.
//   out    = ((256 Dhi + Dlo) + c0) >> 16 with the offsets of the u - 128 trick folded into c0 (taps sum to 256).
// Source tile s of a segment starting at output row y0 is rows y0 - 16 + 16 s ..; output tile T needs source tiles
// T, T + 1, T + 2 (radius <= 15), i.e. slots 1..3 of the window (T - 1 .. T + 2): slot 0 only meets zero taps.
#include "ipx_common.h"

int ipx_sep_mid(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                const uint16_t* kx, const uint16_t* ky, int k, const uint8_t* top, const uint8_t* bot);

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512, TX = 128, PH = 64;
#ifndef IPX_CHAIN_WPS
#define IPX_CHAIN_WPS 4                             // waves per SIMD asked of the compiler (2 blocks per CU at least)
#endif


struct CGeom {
    int h, w, L, r, K;
    int64_t s_row, s_img, d_row, d_img;
    int nstrips, seg_rows, nsegs;
    unsigned nblocks;
    const uint8_t* top;
    const uint8_t* bot;
    int fast32;                                      // row strides small enough for 32-bit in-tile byte offsets
};
struct CTaps { uint8_t kx[32]; uint8_t ky[32]; };

__device__ __forceinline__ const uint8_t* src_row(const CGeom& g, const uint8_t* simg, int y)
{
    return halo_row(simg, g.top, g.bot, y, g.h, g.r, g.s_row);
}

// Workgroup barrier that orders LDS traffic only (__syncthreads() would also drain the prefetch: vmcnt(0)).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ v4i lds16(const uint8_t* p) { return *reinterpret_cast<const v4i*>(p); }

// s_waitcnt vmcnt(N) for a compile-time N (the instruction takes an immediate)
template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N <= 8, "add the immediate");
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

template <int C, int HL>
__global__ __launch_bounds__(kThreads, IPX_CHAIN_WPS) void sep_chain_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                            CGeom g, CTaps taps)
{
    constexpr int KSH = (16 + 2 * HL + 63) / 64;                 // k-steps of the row pass
    constexpr int SW = TX + 2 * HL;                              // source bytes per row of the strip
    constexpr int NCH = SW / 16;                                 // 16-byte chunks per source row (10..16)
    constexpr int PS = HL <= 48 ? 224 : 288;                     // LDS pitch: >= SW, = 32 mod 64 (conflict-free ds_read_b128)
    constexpr int CPR = PS / 16;                                 // chunks per LDS row, padding included
    constexpr int NTOT = PH * CPR / 64;                          // LDS-DMA instructions per phase (64 lanes x 16 B each): 14 or 18
    static_assert(PH * CPR % 64 == 0, "a phase is a whole number of wave-wide DMAs");
    constexpr int NDMA = (NTOT + 7) / 8;                         // ... per wave: NDMA for the first NTOT % 8 waves, NDMA - 1 for the rest
    constexpr int kBuf = PH * PS;                                // bytes per source buffer
    constexpr int NB = 3;                                        // source buffers: phases p + 1 and p + 2 are in flight during phase p
    // the row pass's last k-step reads up to 64 KSH + 112 bytes into a row, i.e. past the pitch of the buffer's last row
    // (finite bytes that only meet zero taps): 256 bytes of slack behind the buffers
    constexpr int PO = 144;                                      // pitch of the output staging tile (64 rows x 128 bytes): 36 dwords
    __shared__ __attribute__((aligned(16))) uint8_t s_all[NB * kBuf + 256 + PH * PO + 64];
    uint8_t* const s_out = s_all + NB * kBuf + 256;
    uint8_t* const s_taps = s_out + PH * PO;

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, gq = lane >> 4;
    const int r = g.r, K = g.K;

    if (tid < 64) s_taps[tid] = tid < 32 ? taps.kx[tid] : taps.ky[tid - 32];

    // block -> (image, row segment, 128-byte strip); neighbouring strips are consecutive items of one XCD (shared halos in L2)
    const unsigned blk = xcd_remap(blockIdx.x, g.nblocks);
    const int strip = (int)(blk % (unsigned)g.nstrips);
    const unsigned rest = blk / (unsigned)g.nstrips;
    const int seg = (int)(rest % (unsigned)g.nsegs);
    const unsigned img = rest / (unsigned)g.nsegs;
    const int x0 = strip * TX, y0 = seg * g.seg_rows;
    const int y1 = y0 + g.seg_rows < g.h ? y0 + g.seg_rows : g.h;
    const int nT = (y1 - y0 + 15) >> 4;                          // output tiles of 16 rows
    const int nS = nT + 2;                                       // source tiles
    const int nP = (nS + 3) >> 2;                                // phases of 4 source tiles
    const uint8_t* const simg = src + (int64_t)img * g.s_img;
    const int xs0 = x0 - HL;
    const bool xin = xs0 >= 0 && xs0 + SW <= g.L;                // block-uniform: the strip's source window lies inside the row

    // ---- source staging: LDS-DMA ------------------------------------------------------------------------------
    // The 64 x PS bytes of a phase are chunk q = row * CPR + chunk of the LDS image; wave w's DMA j moves chunks
    // 128 w' ... : q = 64 (NDMA wave + j) + lane, each lane giving its own global address (row and column clamped into the
    // image; pitch padding and rows past the phase re-load a clamped chunk nobody reads).  global_load_lds writes LDS
    // directly: no registers hold the prefetch, two phases are in flight, and -- since no register of the main loop
    // waits for a load -- the only vector-memory waits are the explicit counted ones below (with register prefetch the
    // compiler's own vmcnt bookkeeping collapsed to vmcnt(0) at every staging, a full drain of the store queue per phase).
    const int nbig = NTOT % 8 ? NTOT % 8 : 8;                     // waves with NDMA instructions
    const int ndma = wave < nbig ? NDMA : NDMA - 1;               // this wave's count (wave-uniform)
    const int dma0 = wave < nbig ? NDMA * wave : NDMA * nbig + (NDMA - 1) * (wave - nbig);   // its first instruction of the phase
    int drow[NDMA], dcol[NDMA];
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const int q = 64 * (dma0 + j) + lane;
        const int rr = q / CPR, cc = q - rr * CPR;
        drow[j] = rr < PH ? rr : PH - 1;
        const int gx = xs0 + 16 * (cc < NCH ? cc : NCH - 1);
        dcol[j] = gx < 0 ? 0 : gx > g.L - 16 ? g.L - 16 : gx;     // (L >= 16: checked on the host)
    }
    auto dma = [&](int p) {
        const int ys0 = y0 - 16 + PH * p;
        uint8_t* const buf = s_all + (p % NB) * kBuf + 1024 * dma0;
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            if (j >= ndma) break;                                 // wave-uniform
            int ys = ys0 + drow[j];
            ys = ys < 0 ? 0 : ys > g.h - 1 ? g.h - 1 : ys;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(simg + (int64_t)ys * g.s_row + dcol[j]),
                                             (void __attribute__((address_space(3)))*)(buf + 1024 * j), 16, 0, 0);
        }
    };
    auto interior = [&](int p) {
        const int ys0 = y0 - 16 + PH * p;
        return xin && ys0 >= 0 && ys0 + PH <= g.h;               // block-uniform
    };
    // Phases at the image's edges: rows from src_row (REFLECT_101 or the caller's halo rows; rows farther out than the
    // radius only meet zero taps and are clamped); a strip that crosses the left / right edge gathers its bytes one by
    // one with REFLECT_101 in pixel space.  Written over what the DMA brought (a few per cent of all phases).
    auto stage_edge = [&](int p, uint8_t* buf) {
        const int lrow = tid >> 4, lch = tid & 15;
        if (lch >= NCH) return;
        const int ys0 = y0 - 16 + PH * p;
        const int gx = xs0 + 16 * lch;
#pragma unroll 1
        for (int i = 0; i < PH / 32; ++i) {
            int ys = ys0 + lrow + 32 * i;
            ys = ys < -r ? -r : ys > g.h - 1 + r ? g.h - 1 + r : ys;
            const uint8_t* rp = src_row(g, simg, ys);
            u32x4 v;
            if (xin) {
                v = ld16(rp + gx);
            } else {
#pragma unroll 1
                for (int wd = 0; wd < 4; ++wd) {
                    uint32_t word = 0;
#pragma unroll 1
                    for (int b = 0; b < 4; ++b) {
                        const int col = gx + 4 * wd + b;
                        int px = col >= 0 ? col / C : -((-col + C - 1) / C);
                        const int chn = col - px * C;
                        px = px < -r ? -r : px > g.w - 1 + r ? g.w - 1 + r : px;   // farther out: never used
                        word |= (uint32_t)rp[reflect101(px, g.w) * C + chn] << (8 * b);
                    }
                    v[wd] = word;
                }
            }
            *reinterpret_cast<u32x4*>(buf + (lrow + 32 * i) * PS + 16 * lch) = v;
        }
    };

    dma(0);
    dma(1);
    lds_barrier();                                                // s_taps (the DMAs stay in flight: no vmcnt wait)

    // ---- Toeplitz operands (registers, the same for every tile) --------------------------------------------------
    v4i bh[KSH], bv[4];
    {
        const int n = m;                                          // B column = lane & 15
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
            uint32_t wds[4] = {0, 0, 0, 0};
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int k = 64 * ks + 16 * gq + jj;             // byte of the wave's source window (starts at x - HL)
                const int d = k - HL - n + r * C;                 // = j * C for tap j
                uint32_t t = 0;
                if (d >= 0 && d % C == 0 && d / C < K) t = s_taps[d / C];
                wds[jj >> 2] |= t << (8 * (jj & 3));
            }
            bh[ks] = (v4i){(int)wds[0], (int)wds[1], (int)wds[2], (int)wds[3]};
        }
        // column pass: the register window keeps source tile S in dword S & 3 (no register moves when the window
        // slides); with S = 4 p + tt the newest tile sits in dword tt, the oldest (S - 3) in dword (tt + 1) & 3: one
        // Toeplitz operand per tt, the k-slots of dword jt meeting the taps of window position (jt - tt - 1) & 3
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            uint32_t wds[4] = {0, 0, 0, 0};
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int wp = ((jj >> 2) - tt - 1) & 3;          // window position of the tile in this dword (0 = oldest)
                const int row = 16 * wp + 4 * gq + (jj & 3);      // window row of k-slot (gq, jj)
                const int i = row - 32 - n + r;                   // output row n sits at window row 32 + n
                uint32_t t = 0;
                if (i >= 0 && i < K) t = s_taps[32 + i];
                wds[jj >> 2] |= t << (8 * (jj & 3));
            }
            bv[tt] = (v4i){(int)wds[0], (int)wds[1], (int)wds[2], (int)wds[3]};
        }
    }

    // Output: the 4 tiles a wave produces per phase (64 rows x 16 columns, 4 bytes per lane and tile) are staged in an
    // LDS tile of 64 rows x 128 bytes and leave as whole 128-byte rows, 16 bytes per lane.  (4-byte stores straight from
    // the accumulator layout wrote 16-byte pieces of 16 different rows per instruction: WRITE_SIZE 3.0x the output.)
    // The store goes through a buffer resource that covers this block's part of the image: a lane whose 16 bytes lie
    // outside (rows before the first / past the last output row, columns past the row) computes an offset beyond the
    // resource and the hardware drops its store -- the instruction itself is unconditional: exactly one per phase.
    uint8_t* const dblk = dst + (int64_t)img * g.d_img + (int64_t)y0 * g.d_row + x0;
    const int wcols = g.L - x0 < TX ? g.L - x0 : TX;              // valid byte columns of this strip
    const unsigned span = (unsigned)((y1 - y0 - 1) * (int)g.d_row + wcols);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(dblk, 0, (int)span, 0x00020000);
    const int orow = tid >> 3, occ = tid & 7;                     // this thread's 16-byte chunk of the staging tile
    uint8_t* const w_out = s_out + m * PO + 16 * wave + 4 * gq;   // + 16 tt PO: row 16 tt + m, columns 16 wave + 4 gq ..
    const uint8_t* const r_out = s_out + orow * PO + 16 * occ;
    const bool ragged = (wcols & 15) != 0;                        // block-uniform, rare: the strip ends inside a 16-byte chunk
    // columns past the strip's valid width: an offset that no row offset brings back inside the resource (span < 2^30)
    const unsigned lane_off = 16 * occ + 16 <= wcols ? 16u * (unsigned)occ : 0x80000000u;
    uint32_t whi[4] = {0, 0, 0, 0}, wlo[4] = {0, 0, 0, 0};        // register window: high / low bytes of the last 4 source tiles

    for (int p = 0; p < nP; ++p) {
        uint8_t* const buf = s_all + (p % NB) * kBuf;
        // This wave's vector-memory queue, oldest first: DMA(p), store of phase p - 2, DMA(p + 1), store of phase p - 1.
        // Wait until DMA(p) has landed: nw younger operations may still be outstanding (its ndma DMAs of phase p + 1,
        // plus one store in phase 1 and two from phase 2 on; the byte stores of a ragged strip and the loads of an edge phase only add
        // younger ones, which makes the same count stricter, never wrong).
        const int nw = ndma + (p < 2 ? p : 2);                    // wave-uniform
        if (nw >= NDMA + 2) wait_vmcnt<NDMA + 2>();
        else if (nw == NDMA + 1) wait_vmcnt<NDMA + 1>();
        else if (nw == NDMA) wait_vmcnt<NDMA>();
        else wait_vmcnt<NDMA - 1>();
        if (!interior(p)) {
            __builtin_amdgcn_s_barrier();                         // every wave's DMA(p) has landed before it is overwritten
            stage_edge(p, buf);
        }
        lds_barrier();              // DMA(p) of every wave visible; every wave has left phase p - 1: its buffers are free
        dma(p + 2);                 // (past the last phase: clamped rows, never read)
        __builtin_amdgcn_sched_barrier(0);                        // the counted waits rely on this order: DMA, then the store

        const uint8_t* const a_h = buf + m * PS + 16 * gq + 16 * wave;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            // ---- row pass of source tile S = 4 p + tt: 16 rows x this wave's 16 columns (A = u - 128 as signed bytes)
            v4i acc = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks)
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(lds16(a_h + 16 * tt * PS + 64 * ks) ^ (v4i)(int)0x80808080u, bh[ks], acc, 0, 0, 0);
            // acc[i] = row 4 gq + i, column m; Q8.8 value h = acc + 2^15.  (h >> 8) - 128 is byte 1 of acc itself;
            // (h & 255) - 128 = byte 0 ^ 0x80
            const uint32_t h0 = (uint32_t)acc[0], h1 = (uint32_t)acc[1], h2 = (uint32_t)acc[2], h3 = (uint32_t)acc[3];
            whi[tt] = __builtin_amdgcn_perm(h1, h0, 0x0c0c0501u) | __builtin_amdgcn_perm(h3, h2, 0x05010c0cu);
            wlo[tt] = (__builtin_amdgcn_perm(h1, h0, 0x0c0c0400u) | __builtin_amdgcn_perm(h3, h2, 0x04000c0cu)) ^ 0x80808080u;
            // ---- column pass of output tile T = S - 2 (tiles before the first / past the last: computed, never stored)
            constexpr int c0 = 32768 * 256 + 32768 + 32768;       // rounding and the two +2^15 offsets, via the accumulator
            const v4i z = {0, 0, 0, 0}, zc = {c0, c0, c0, c0};
            const v4i ahi = {(int)whi[0], (int)whi[1], (int)whi[2], (int)whi[3]}, alo = {(int)wlo[0], (int)wlo[1], (int)wlo[2], (int)wlo[3]};
            const v4i dhi = __builtin_amdgcn_mfma_i32_16x16x64_i8(ahi, bv[tt], z, 0, 0, 0);
            const v4i dlo = __builtin_amdgcn_mfma_i32_16x16x64_i8(alo, bv[tt], zc, 0, 0, 0);
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = ((uint32_t)dhi[i] << 8) + (uint32_t)dlo[i];
            // o[i] bits 16..23 = output byte of column 16 wave + 4 gq + i, row 16 T + m
            *reinterpret_cast<uint32_t*>(w_out + 16 * tt * PO) =
                __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0602u) | __builtin_amdgcn_perm(o[3], o[2], 0x06020c0cu);
        }
        lds_barrier();
        // staging row orow is output row 16 (4 p - 2) + orow of the segment; a negative row wraps to an offset beyond the span
        const u32x4 v = *reinterpret_cast<const u32x4*>(r_out);
        const int yr = 16 * (4 * p - 2) + orow;
        const unsigned off = (unsigned)(yr * (int)g.d_row) + lane_off;
        __builtin_amdgcn_raw_buffer_store_b128(v, drs, off, 0, 2);
        if (ragged) {                                             // the strip's last, partial chunk: byte by byte
            asm volatile("" ::: "memory");                        // (keeps this rare branch a branch)
            if (16 * occ < wcols && 16 * occ + 16 > wcols && yr >= 0 && yr < y1 - y0)
                for (int b = 0; 16 * occ + b < wcols; ++b)
                    dblk[(int64_t)yr * g.d_row + 16 * occ + b] = (uint8_t)(v[b >> 2] >> (8 * (b & 3)));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the DMAs issued past the last phase land before LDS is released
}

template <int C>
void launch_c(hipStream_t st, const uint8_t* src, uint8_t* dst, const CGeom& g, const CTaps& t)
{
    const int hl = (g.r * C + 15) & ~15;
    const dim3 grid(g.nblocks), block(kThreads);
    switch (hl) {
        case 16: hipLaunchKernelGGL((sep_chain_kernel<C, 16>), grid, block, 0, st, src, dst, g, t); break;
        case 32: hipLaunchKernelGGL((sep_chain_kernel<C, 32>), grid, block, 0, st, src, dst, g, t); break;
        case 48: hipLaunchKernelGGL((sep_chain_kernel<C, 48>), grid, block, 0, st, src, dst, g, t); break;
        default: hipLaunchKernelGGL((sep_chain_kernel<C, 64>), grid, block, 0, st, src, dst, g, t); break;
    }
}

}  // namespace

// Called by ipx_sep_filter_q8_u8 for 9 <= k <= 31 (arguments validated there).
int ipx_sep_chain(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                  int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                  const uint16_t* kx, const uint16_t* ky, int k, const uint8_t* top, const uint8_t* bot)
{
    const char* fn = "ipx_sep_filter_q8_u8";
    bool fits = true;
    for (int i = 0; i < k; ++i) fits = fits && kx[i] <= 127 && ky[i] <= 127;
    // rows shorter than one 16-byte load, or row strides beyond 32-bit in-block offsets: the VALU two-kernel path
    if (!fits || w * c < 16 || d_row >= (1 << 24))
        return ipx_sep_mid(st, src, dst, n, h, w, c, s_row, s_img, d_row, d_img, kx, ky, k, top, bot);
    CGeom g;
    g.h = h; g.w = w; g.L = w * c; g.r = k / 2; g.K = k;
    g.s_row = s_row; g.s_img = s_img; g.d_row = d_row; g.d_img = d_img; g.top = top; g.bot = bot;
    g.fast32 = 1;
    g.nstrips = (g.L + TX - 1) / TX;
    // rows per block: about 2048 blocks in flight order (256 CUs x 4 resident blocks x 2), at least 64 rows each; every
    // segment re-reads 32 rows (two source tiles)
    const int64_t strips = (int64_t)n * g.nstrips;
    int nsegs = (int)((2048 + strips - 1) / strips);
    if (nsegs < 1) nsegs = 1;
    int seg = ((h + nsegs - 1) / nsegs + 15) / 16 * 16;
    if (seg < 64) seg = 64;
    while (seg > 64 && (int64_t)(seg + 16) * d_row >= (1ll << 30)) seg = (seg / 2 + 15) / 16 * 16;   // 32-bit store offsets
    g.seg_rows = seg;
    g.nsegs = (h + seg - 1) / seg;
    const uint64_t nb = (uint64_t)strips * g.nsegs;
    if (nb > 0x7ffffff0ull) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: image too large", fn);
    g.nblocks = (unsigned)nb;
    CTaps t{};
    for (int i = 0; i < k; ++i) { t.kx[i] = (uint8_t)kx[i]; t.ky[i] = (uint8_t)ky[i]; }
    switch (c) {
        case 1: launch_c<1>(st, src, dst, g, t); break;
        case 2: launch_c<2>(st, src, dst, g, t); break;
        case 3: launch_c<3>(st, src, dst, g, t); break;
        default: launch_c<4>(st, src, dst, g, t); break;
    }
    IPX_LAUNCH_CHECK("sep_chain_kernel");
    return IPX_OK;
}

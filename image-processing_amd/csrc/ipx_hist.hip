// ipx_hist.hip -- 256-bin histogram, equalizeHist LUT construction, cv2.equalizeHist.
//
// hist_partial_kernel: each block counts a slice of one image into an LDS histogram laid out
//   [256 bins][32 copies] (32 KiB): lane l always adds to copy (l & 31), i.e. to LDS bank (l & 31), so a
//   wave's 64 ds_add_u32 never collide on a bank whatever the pixel values are -- a constant image
//   (all lanes, one bin) costs the same as white noise.  The 32 copies are folded with a rotated
//   (conflict-free) read and the block stores its 256 partial counts with plain coalesced stores: no
//   global atomics, no zero-fill, bitwise reproducible.
// hist_tail: for images cut into 65 or more blocks, the LAST block to arrive at a ticket (one device-scope atomic per block;
//   the tickets live in the caller's workspace, start at zero and are set back to zero by the block that wins them) sums
//   the partial counts, in two levels (groups of 32 blocks, then <= 32 group sums: one round of loads each), and builds the
//   equalization table with an LDS scan, in float32 exactly as OpenCV does: no second launch between the histogram pass and
//   the table pass.
// hist_finish_kernel: the same sum and table construction as a kernel of its own, one block per image: batches of images
//   of fewer than 65 blocks each (where it measured faster), and ipx_equalize_lut_u8 (a row-partitioned image all-reduces
//   its histogram first).
#include "ipx_common.h"
#include <stdlib.h>
#include <atomic>

namespace {

constexpr int kThreads = 512;                  // 8 waves share one 32 KiB LDS histogram: 4 blocks = 32 waves per CU
constexpr int kUnroll = 4;
constexpr int kChunk = kThreads * 16;
constexpr int kUnit = kChunk * kUnroll;          // 16 KiB of pixels per block step
constexpr int kCopies = 32;

struct HistGeom {
    int rows;
    int64_t len;
    int64_t s_row, s_img;
    unsigned units_per_row, units_per_img;
};

__device__ __forceinline__ void count4(uint32_t* s_hist, uint32_t x, unsigned copy)
{
    atomicAdd(&s_hist[((x & 255u) << 5) | copy], 1u);
    atomicAdd(&s_hist[(((x >> 8) & 255u) << 5) | copy], 1u);
    atomicAdd(&s_hist[(((x >> 16) & 255u) << 5) | copy], 1u);
    atomicAdd(&s_hist[((x >> 24) << 5) | copy], 1u);
}

// ---- in-kernel finish ---------------------------------------------------------------------------------------------
constexpr unsigned kTicketImages = 4096;       // one ticket per image (the only level, or the second of two)
constexpr unsigned kTicketGroups = 128;        // first-level tickets: one per group of kGroup blocks
constexpr size_t kTicketBytes = (size_t)(kTicketImages + kTicketGroups) * sizeof(uint32_t);    // 66 x 256 B, at the
                                               // START of every workspace: the same words whatever the batch geometry
constexpr unsigned kGroup = 32;                // blocks per first-level group; at most kGroup * kGroup blocks per image
constexpr unsigned kTicketMinBlocks = 65;      // images cut into fewer blocks keep the separate finish kernel (see make_tail)

struct HistTail {
    uint32_t* partial;      // [n][blocks per image][256]
    uint32_t* partial2;     // [n][ngroups][256]   (ngroups > 1 only)
    uint32_t* tickets;      // null: partial counts only, the host launches hist_finish_kernel
    uint32_t* hist_out;     // may be null
    uint8_t* lut_out;       // may be null
    unsigned ngroups;       // ceil(blocks per image / kGroup), <= kGroup
};

// Counts that cross blocks (and so, possibly, XCDs, whose L2s are not coherent with each other) move with device-scope
// relaxed atomics: the store is written through to memory (sc1), the load bypasses the caches.  NO device-scope fence
// anywhere: on gfx950 such a fence writes back and invalidates the whole L2 of the XCD (buffer_wbl2 / buffer_inv), and
// with one per wave of every block the histogram pass ran 8 x slower (measured: 327 us instead of 40 us on 128 MiB).
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// All threads of the block call this after storing their counts with st_agent.  True (block-uniform) in the block that
// arrives last of `count` at `ticket`; that block puts the ticket back to zero for the next call.  Order: every wave waits
// until its write-through stores are acknowledged (vmcnt 0), the barrier collects the waves, THEN thread 0 moves the
// ticket; the winner's loads are issued after the barrier that follows the atomic's return.
__device__ __forceinline__ bool arrive_last(uint32_t* ticket, unsigned count, uint32_t* s_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old + 1u == count;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last ? 1u : 0u;
    }
    __syncthreads();
    return *s_flag != 0;
}

// Sum of `count` (<= kGroup) 256-bin count vectors at p (stride 256 words): bin tid & 255, the two halves of the
// 512-thread block take alternate vectors.  Every load of a thread is issued before the first is used -- one memory round
// trip, not count / 16 of them (the winner of a ticket is the only block still running: latency is all that matters).
// The result is valid in threads 0..255.  s_tmp: 256 words of LDS.
__device__ __forceinline__ uint32_t sum_counts(const uint32_t* p, unsigned count, uint32_t* s_tmp, unsigned tid)
{
    const unsigned bin = tid & 255u, half = tid >> 8;
    p += bin;
    uint32_t t[kGroup / 2];
    // unconditional loads (vectors past the end re-read the last one and are not added): a load behind a branch would
    // make the compiler wait for the loads in flight before it writes the zero of the other arm
#pragma unroll
    for (unsigned k = 0; k < kGroup / 2; ++k) {
        const unsigned v = 2 * k + half;
        t[k] = ld_agent(p + (size_t)(v < count ? v : count - 1) * 256);
    }
    uint32_t h = 0;
#pragma unroll
    for (unsigned k = 0; k < kGroup / 2; ++k) h += 2 * k + half < count ? t[k] : 0u;
    if (half) s_tmp[bin] = h;
    __syncthreads();
    if (!half) h += s_tmp[bin];
    return h;
}

// OpenCV histogram.cpp equalizeHist(): i0 = first non-zero bin; hist[i0]==total -> constant image maps to i0;
// scale = 255.f/(total-hist[i0]); lut[i0]=0; lut[i>i0] = saturate_cast<uchar>(cvRound(float(sum_i)*scale)).
// Called by exactly 256 threads (tid = bin, h = its count) that share s_scan (2 x 256 words) and s_i0; the block's other
// threads have exited (exited waves no longer count at the barrier).
__device__ __forceinline__ void finish_image(uint32_t h, unsigned tid, unsigned img, uint32_t* __restrict__ hist_out,
                                             uint8_t* __restrict__ lut_out, uint32_t (*s_scan)[256], int* s_i0)
{
    if (hist_out) hist_out[(size_t)img * 256 + tid] = h;
    if (!lut_out) return;
    if (tid == 0) *s_i0 = 256;
    s_scan[0][tid] = h;
    __syncthreads();
    if (h != 0) atomicMin(s_i0, (int)tid);
    int cur = 0;                                             // Hillis-Steele inclusive scan over 256 bins
#pragma unroll
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t v = s_scan[cur][tid];
        if (tid >= (unsigned)off) v += s_scan[cur][tid - off];
        s_scan[cur ^ 1][tid] = v;
        cur ^= 1;
        __syncthreads();
    }
    const uint32_t incl = s_scan[cur][tid];                  // sum of bins 0..tid
    const uint32_t total = s_scan[cur][255];
    const int i0 = *s_i0;
    uint32_t out = 0;
    // The raw counts were overwritten by the scan; hist[i0] == incl(i0) because every bin below i0 is empty.
    if (i0 < 256) {
        const uint32_t hist_i0 = s_scan[cur][i0];
        if (hist_i0 == total) {
            out = (uint32_t)i0;                              // single-valued image: dst.setTo(i0)
        } else if ((int)tid > i0) {
            const float scale = __fdiv_rn(255.f, (float)(int)(total - hist_i0));
            const float v = __builtin_rintf(__fmul_rn((float)(int)(incl - hist_i0), scale));
            out = v < 0.f ? 0u : v > 255.f ? 255u : (uint32_t)v;
        }
    }
    lut_out[(size_t)img * 256 + tid] = (uint8_t)out;
}

// End of a counting block (kThreads threads, all of them, after the barrier that follows the counting): fold the 32 copies
// of the LDS histogram, store the block's counts and, in the last block of the image to get here, finish the image.
__device__ __forceinline__ void hist_tail(uint32_t* s_hist, unsigned img, const HistTail& t)
{
    __shared__ uint32_t s_flag;
    __shared__ int s_i0;
    const unsigned tid = threadIdx.x, nparts = gridDim.x, b = blockIdx.x;
    uint32_t h = 0;
    // lane reads copy (j + tid) % 32 -> bank (j + tid) % 32: conflict-free
    if (tid < 256) {
#pragma unroll
        for (int j = 0; j < kCopies; ++j) h += s_hist[(tid << 5) | ((j + tid) & 31u)];
    }
    if (!t.tickets) {                                        // hist_finish_kernel follows (the launch boundary publishes these)
        if (tid < 256) t.partial[((size_t)img * nparts + b) * 256 + tid] = h;
        return;
    }
    // level 1: groups of kGroup consecutive blocks; the last of a group to arrive sums the group
    if (tid < 256) st_agent(&t.partial[((size_t)img * nparts + b) * 256 + tid], h);
    const unsigned grp = b / kGroup, first = grp * kGroup;
    const unsigned cnt = nparts - first < kGroup ? nparts - first : kGroup;
    if (!arrive_last(t.tickets + kTicketImages + img * t.ngroups + grp, cnt, &s_flag)) return;
    h = sum_counts(t.partial + ((size_t)img * nparts + first) * 256, cnt, s_hist, tid);
    // level 2: the last group of the image to be summed sums the group sums
    if (tid < 256) st_agent(&t.partial2[((size_t)img * t.ngroups + grp) * 256 + tid], h);
    if (!arrive_last(t.tickets + img, t.ngroups, &s_flag)) return;
    h = sum_counts(t.partial2 + (size_t)img * t.ngroups * 256, t.ngroups, s_hist, tid);
    if (tid >= 256) return;
    finish_image(h, tid, img, t.hist_out, t.lut_out, (uint32_t (*)[256])(s_hist + 256), &s_i0);
}

template <bool NT>
__global__ __launch_bounds__(kThreads) void hist_partial_kernel(const uint8_t* __restrict__ src, HistGeom g, HistTail tail)
{
    __shared__ uint32_t s_hist[256 * kCopies];
    const unsigned tid = threadIdx.x, copy = tid & 31u;
#pragma unroll
    for (int i = 0; i < 256 * kCopies / kThreads; ++i) s_hist[i * kThreads + tid] = 0;
    __syncthreads();

    const unsigned img = blockIdx.y;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    // each block takes a CONTIGUOUS range of units (streams faster than striding over the image)
    unsigned u_begin = (unsigned)((uint64_t)g.units_per_img * blockIdx.x / gridDim.x);
    const unsigned u_end = (unsigned)((uint64_t)g.units_per_img * (blockIdx.x + 1) / gridDim.x);
    // software pipeline: the loads of unit u + 1 are issued before unit u is counted (a block owns up to 8 units of a
    // large single image; without the overlap every unit cost a full memory round trip: 3.3 TB/s on a 128 MiB shard)
    auto unit_ptr = [&](unsigned u, int64_t& col0) {
        const unsigned row = u / g.units_per_row;
        col0 = (int64_t)(u - row * g.units_per_row) * kUnit + (int64_t)tid * 16;
        return simg + (int64_t)row * g.s_row;
    };
    auto issue = [&](unsigned u, u32x4 (&v)[kUnroll]) {
        int64_t col0;
        const uint8_t* sp = unit_ptr(u, col0);
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            // unconditional (lanes past the row's end re-load its last 16 bytes; rows are >= 16 bytes here): no load sits
            // behind a branch, so the waits the compiler places before count() are counted ones and the next unit's
            // loads stay in flight
            const int64_t c = col0 + (int64_t)j * kChunk;
            v[j] = ld16x<NT>(sp + (c + 16 <= g.len ? c : g.len - 16));
        }
    };
    auto count = [&](unsigned u, const u32x4 (&v)[kUnroll]) {
        int64_t col0;
        const uint8_t* sp = unit_ptr(u, col0);
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            const int64_t c = col0 + (int64_t)j * kChunk;
            if (c + 16 <= g.len) {
#pragma unroll
                for (int k = 0; k < 4; ++k) count4(s_hist, v[j][k], copy);
            } else if (c < g.len) {
                for (int64_t b = c; b < g.len; ++b) atomicAdd(&s_hist[((uint32_t)sp[b] << 5) | copy], 1u);
            }
        }
    };
    if (g.len < 16) {                                            // rows shorter than one vector load: byte by byte
        for (unsigned u = u_begin; u < u_end; ++u) {
            int64_t col0;
            const uint8_t* sp = unit_ptr(u, col0);
            if (tid == 0)
                for (int64_t b = 0; b < g.len; ++b) atomicAdd(&s_hist[((uint32_t)sp[b] << 5)], 1u);
        }
        u_begin = u_end;
    }
    u32x4 va[kUnroll], vb[kUnroll];
    if (u_begin < u_end) issue(u_begin, va);
    for (unsigned u = u_begin; u < u_end; u += 2) {
        if (u + 1 < u_end) issue(u + 1, vb);
        count(u, va);
        if (u + 1 < u_end) {
            if (u + 2 < u_end) issue(u + 2, va);
            count(u + 1, vb);
        }
    }
    __syncthreads();
    hist_tail(s_hist, img, tail);
}


// Fused cvtColor(BGR2GRAY) + histogram: reads BGR(A), writes the gray image (default cache policy: the table pass
// re-reads it right away) and counts it.  Same LDS layout and partial-sum output as hist_partial_kernel.
struct GrayHistGeom {
    int rows;
    int64_t len;                       // pixels per row (after collapsing dense images to one long row)
    int64_t s_row, s_img, g_row, g_img;
    unsigned units_per_row, units_per_img;
};
constexpr int kGUnit = kChunk * 2;     // pixels per block step: 2 x (3|4) x 16 B loads in flight per lane

template <int C, bool NT>
__global__ __launch_bounds__(kThreads) void gray_hist_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ gray,
                                                             GrayHistGeom g, HistTail tail)
{
    __shared__ uint32_t s_hist[256 * kCopies];
    __shared__ __attribute__((aligned(16))) uint8_t s_tr[C == 3 ? (kThreads / 64) * 3072 : 16];   // transpose_group3
    const unsigned tid = threadIdx.x, copy = tid & 31u;
    const int lane = (int)(tid & 63u), wave = (int)(tid >> 6);
    uint8_t* wlds = s_tr + (C == 3 ? wave * 3072 : 0);
#pragma unroll
    for (int i = 0; i < 256 * kCopies / kThreads; ++i) s_hist[i * kThreads + tid] = 0;
    __syncthreads();
    const unsigned img = blockIdx.y;
    const uint8_t* simg = src + (int64_t)img * g.s_img;
    uint8_t* gimg = gray + (int64_t)img * g.g_img;
    const unsigned u_begin = (unsigned)((uint64_t)g.units_per_img * blockIdx.x / gridDim.x);
    const unsigned u_end = (unsigned)((uint64_t)g.units_per_img * (blockIdx.x + 1) / gridDim.x);
    for (unsigned u = u_begin; u < u_end; ++u) {
        const unsigned row = u / g.units_per_row;
        const int64_t gx0 = (int64_t)(u - row * g.units_per_row) * kGUnit + (int64_t)wave * 1024;   // wave's pixel group
        const uint8_t* sp = simg + (int64_t)row * g.s_row;
        uint8_t* gp = gimg + (int64_t)row * g.g_row;
        PxVec<C> p[2];
        u32x4 raw[2][3];
        bool full[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t gx = gx0 + (int64_t)j * kChunk, x = gx + 16 * lane;
            full[j] = C == 3 && gx + 1024 <= g.len;       // wave-uniform
            if (full[j]) load_group3_nt(raw[j], sp + gx * 3, lane);
            else if (x + 16 <= g.len) load_px16<C, NT>(p[j], sp + x * C);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t x = gx0 + (int64_t)j * kChunk + 16 * lane;
            if (C == 3 && full[j]) {
                PxVec<3> q;
                transpose_group3(q, raw[j], wlds, lane);
                __builtin_memcpy(&p[j], &q, sizeof(q));
            }
            if (x + 16 <= g.len) {
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t word = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int px = 4 * k + i;
                        word |= gray_q15(px_byte<C>(p[j], px * C), px_byte<C>(p[j], px * C + 1), px_byte<C>(p[j], px * C + 2)) << (8 * i);
                    }
                    o[k] = word;
                    count4(s_hist, word, copy);
                }
                st16(gp + x, o);
            } else if (x < g.len) {
                for (int64_t q = x; q < g.len; ++q) {
                    const uint8_t* s3 = sp + q * C;
                    const uint32_t y = gray_q15(s3[0], s3[1], s3[2]);
                    gp[q] = (uint8_t)y;
                    atomicAdd(&s_hist[(y << 5) | copy], 1u);
                }
            }
        }
    }
    __syncthreads();
    hist_tail(s_hist, img, tail);
}

// One block per image: hist[bin] = sum over that image's partial blocks (or a given histogram); optional equalization LUT.
constexpr int kFinishGroups = 4;                             // 4 x 256 threads share the partial-sum latency chain
__global__ __launch_bounds__(256 * kFinishGroups) void hist_finish_kernel(const uint32_t* __restrict__ partial, unsigned nparts,
                                                               const uint32_t* __restrict__ hist_in,
                                                               uint32_t* __restrict__ hist_out,
                                                               uint8_t* __restrict__ lut_out)
{
    __shared__ uint32_t s_scan[2][256];
    __shared__ uint32_t s_part[kFinishGroups][256];
    __shared__ int s_i0;
    const unsigned tid = threadIdx.x & 255u, grp = threadIdx.x >> 8, img = blockIdx.x;   // (own block size: 1024)
    uint32_t h = 0;
    if (partial) {
        const uint32_t* p = partial + (size_t)img * nparts * 256 + tid;
        unsigned b = grp;
        for (; b + 7 * kFinishGroups < nparts; b += 8 * kFinishGroups) {
            uint32_t t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = p[(size_t)(b + k * kFinishGroups) * 256];
#pragma unroll
            for (int k = 0; k < 8; ++k) h += t[k];
        }
        for (; b < nparts; b += kFinishGroups) h += p[(size_t)b * 256];
        s_part[grp][tid] = h;
        __syncthreads();
        h = 0;
#pragma unroll
        for (int k = 0; k < kFinishGroups; ++k) h += s_part[k][tid];
    } else {
        h = hist_in[(size_t)img * 256 + tid];
    }
    if (grp != 0) return;                                    // groups 1.. only helped with the partial sums
    finish_image(h, tid, img, hist_out, lut_out, s_scan, &s_i0);
}

HistGeom make_geom(int h, int w, int64_t s_row, int64_t s_img)
{
    HistGeom g;
    g.rows = h; g.len = w; g.s_row = s_row; g.s_img = s_img;
    if (s_row == w) { g.rows = 1; g.len = (int64_t)w * h; }
    g.units_per_row = ipx_div_up((uint64_t)g.len, kUnit);
    g.units_per_img = (unsigned)g.rows * g.units_per_row;
    return g;
}

// blocks per image: enough blocks to fill the chip (LDS admits 4-5 blocks per CU) in one round
unsigned blocks_per_image(int n, int h, int w, unsigned slots = 1024)
{
    const uint64_t bytes = (uint64_t)h * (uint64_t)w;
    // at least 64 KiB of pixels per block, so that zeroing and folding the 32 KiB LDS histogram stays in the noise -- unless
    // that leaves CUs without a block: one 4K frame is 127 such blocks on 256 CUs; 32 KiB each: 20.6 -> 16.0 us for
    // cvtColor + histogram of one frame (scripts/dev/onetune.py)
    uint64_t grain = (uint64_t)n * bytes / 256 / 16384 * 16384;
    grain = grain < 16384 ? 16384 : grain > 65536 ? 65536 : grain;
#ifdef IPX_DEV
    if (const char* e = getenv("IPX_HIST_SLOTS")) { if (atoi(e) > 0) slots = (unsigned)atoi(e); }     // (re-read per call: sweeps)
    if (const char* e = getenv("IPX_HIST_GRAIN")) { if (atoi(e) > 0) grain = (uint64_t)atoi(e); }
#endif
    uint64_t by_work = (bytes + grain - 1) / grain;
    uint64_t by_chip = slots / (uint64_t)n;                      // 256 CUs x resident blocks per CU: ONE round (rounding up
                                                                 // put e.g. 24 images x 22 blocks = 528 blocks on 512 slots)
    uint64_t b = by_work < by_chip ? by_work : by_chip;
    return (unsigned)(b < 1 ? 1 : b);
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

unsigned groups_of(unsigned nparts) { return ipx_div_up(nparts, kGroup); }

// [tickets][partial counts][group sums][histograms][tables].  The tickets come first so that they are the same words for
// every batch geometry a workspace is used with: they are zero between calls, everything behind them is overwritten.
struct Workspace {
    uint32_t* tickets; uint32_t* partial; uint32_t* partial2; uint32_t* hist; uint8_t* lut;
};

size_t workspace_layout(int n, int h, int w, Workspace* ws, uint8_t* base)
{
    const unsigned bpi = blocks_per_image(n, h, w);
    const size_t p = align256((size_t)n * bpi * 256 * sizeof(uint32_t));
    const size_t p2 = bpi >= kTicketMinBlocks ? align256((size_t)n * groups_of(bpi) * 256 * sizeof(uint32_t)) : 0;
    const size_t hs = align256((size_t)n * 256 * sizeof(uint32_t));
    const size_t ls = align256((size_t)n * 256);
    if (ws) {
        ws->tickets = (uint32_t*)base;
        ws->partial = (uint32_t*)(base + kTicketBytes);
        ws->partial2 = (uint32_t*)(base + kTicketBytes + p);
        ws->hist = (uint32_t*)(base + kTicketBytes + p + p2);
        ws->lut = base + kTicketBytes + p + p2 + hs;
    }
    return kTicketBytes + p + p2 + hs + ls;
}

// The in-kernel finish for a launch of n x nparts blocks, or tickets = null: the host launches hist_finish_kernel afterwards.
// Which one is a measured choice (scripts/dev/histeq.py, same box, A/B): every block of a ticketed launch pays two serial
// device-scope round trips at its end (write-through store acknowledged, then the atomic) and the winner two more per
// level -- about 8 us whatever the geometry; the separate kernel costs its launch edge plus a sum whose length grows with
// the blocks per image.  One image in 1024 blocks (a 128 MiB row partition): tickets 4 us faster; one 4K frame in 127
// blocks: 2-3 us faster; 24 frames of 42 blocks each: tickets 2.7 us SLOWER.  Hence: tickets from 65 blocks per image up.
HistTail make_tail(const Workspace& ws, int n, unsigned nparts, uint32_t* hist_out, uint8_t* lut_out)
{
    HistTail t;
    t.partial = ws.partial; t.partial2 = ws.partial2; t.hist_out = hist_out; t.lut_out = lut_out;
    t.ngroups = groups_of(nparts);
    bool covered = nparts >= kTicketMinBlocks && t.ngroups <= kGroup && (unsigned)n <= kTicketImages &&
                   (uint64_t)n * t.ngroups <= kTicketGroups;
#ifdef IPX_DEV
    static const bool env_off = [] { const char* e = getenv("IPX_HIST_NOTICKET"); return e && atoi(e) != 0; }();   // A/B: separate finish
    if (env_off) covered = false;
#endif
    t.tickets = covered ? ws.tickets : nullptr;
    return t;
}

// after a counting kernel launched with `t`: the separate finish where the in-kernel one was not possible
int finish_if_needed(hipStream_t st, const HistTail& t, int n, unsigned nparts)
{
    if (t.tickets) return IPX_OK;
    hipLaunchKernelGGL(hist_finish_kernel, dim3((unsigned)n), dim3(256 * kFinishGroups), 0, st, t.partial, nparts,
                       (const uint32_t*)nullptr, t.hist_out, t.lut_out);
    IPX_LAUNCH_CHECK("hist_finish_kernel");
    return IPX_OK;
}

int check_hist_args(const char* fn, const uint8_t* src, int n, int h, int w, int64_t s_row, int64_t s_img)
{
    if (int rc = ipx_check_image_args(fn, src, src, n, h, w, 1, 1, s_row, s_img, s_row, s_img)) return rc;
    if ((uint64_t)h * (uint64_t)w > 0x7fffffffull)
        return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: %d x %d pixels exceed int32 counts (OpenCV's own limit)", fn, h, w);
    if (n > 65535) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: n=%d > 65535 images per call", fn, n);
    return IPX_OK;
}

// counting pass + finish: histograms to hist_out and/or tables to lut_out (either may be null)
int launch_hist(hipStream_t st, const uint8_t* src, int n, int h, int w, int64_t s_row, int64_t s_img,
                const Workspace& ws, uint32_t* hist_out, uint8_t* lut_out, uint64_t nt_threshold = kNtHistThenLut)
{
    HistGeom g = make_geom(h, w, s_row, s_img);
    unsigned bpi = blocks_per_image(n, h, w);
    if (bpi > g.units_per_img) bpi = g.units_per_img;     // strided rows: fewer units than bytes/64K suggests
    const HistTail t = make_tail(ws, n, bpi, hist_out, lut_out);
    // the histogram pass reads what the table pass re-reads right after: keep small batches in the Infinity Cache
    if (ipx_use_nt(2ull * (uint64_t)n * h * w, nt_threshold))
        hipLaunchKernelGGL(hist_partial_kernel<true>, dim3(bpi, (unsigned)n), dim3(kThreads), 0, st, src, g, t);
    else
        hipLaunchKernelGGL(hist_partial_kernel<false>, dim3(bpi, (unsigned)n), dim3(kThreads), 0, st, src, g, t);
    IPX_LAUNCH_CHECK("hist_partial_kernel");
    return finish_if_needed(st, t, n, bpi);
}

// hist -> table -> table pass around a workspace laid out by workspace_layout (shared by the allocating and the
// no-allocation entry points)
int equalize_with_workspace(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                            int64_t s_row, int64_t s_img, int64_t d_row, int64_t d_img,
                            uint32_t* hist_out_dev, uint8_t* lut_out_dev, uint8_t* workspace)
{
    Workspace ws;
    workspace_layout(n, h, w, &ws, workspace);
    uint8_t* lut = lut_out_dev ? lut_out_dev : ws.lut;
    // (tables only: the caller's own table pass, possibly of several tables, reads the batch next -- leave it in the cache longer)
    if (int rc = launch_hist(st, src, n, h, w, s_row, s_img, ws, hist_out_dev, lut, dst ? kNtHistThenLut : kNtHistOnly)) return rc;
    if (!dst) return IPX_OK;                                // histograms and tables only: the caller applies the tables
    // the table pass (default cache policy up to kNtHistThenLut bytes: what the histogram pass read is still in the Infinity Cache)
    return ipx_lut_dev_u8((void*)st, src, dst, n, h, w, 1, s_row, s_img, d_row, d_img, lut, 256);
}

// A workspace for one call: the caller's (tickets already zero: ipx_hist_workspace_init) or one from the stream-ordered
// allocator, whose tickets are cleared here.
int take_workspace(const char* fn, hipStream_t st, size_t need, void** workspace, size_t workspace_bytes, void** own)
{
    *own = nullptr;
    if (*workspace) {
        if (workspace_bytes < need) return ipx_fail(IPX_ERR_INVALID, "%s: workspace %zu < %zu bytes needed", fn, workspace_bytes, need);
        if ((uintptr_t)*workspace & 3u) return ipx_fail(IPX_ERR_INVALID, "%s: workspace %p is not 4-byte aligned", fn, *workspace);
        return IPX_OK;
    }
    IPX_HIP(hipMallocAsync(own, need, st));
    hipError_t e = hipMemsetAsync(*own, 0, kTicketBytes, st);
    if (e != hipSuccess) {
        (void)hipFreeAsync(*own, st);
        *own = nullptr;
        return ipx_fail(IPX_ERR_HIP, "%s: hipMemsetAsync failed: %s", fn, hipGetErrorString(e));
    }
    *workspace = *own;
    return IPX_OK;
}

int release_workspace(hipStream_t st, void* own, int rc)
{
    if (own) {
        hipError_t e = hipFreeAsync(own, st);
        if (e != hipSuccess && rc == IPX_OK) rc = ipx_fail(IPX_ERR_HIP, "hipFreeAsync failed: %s", hipGetErrorString(e));
    }
    return rc;
}

// ---- scratch of the no-allocation entry point ---------------------------------------------------------------------
// One array per GPU, held until the process ends.  `scratch` is a host-side stand-in for the symbol, NOT a device
// address: host code only ever uses the pointer hipGetSymbolAddress returned for the device the call's stream belongs to.
constexpr size_t kScratchBytes = IPX_NOALLOC_SCRATCH_BYTES;           // partial counts (<= 1 MiB) + histograms + tables of <= 256 images
constexpr int kMaxDevices = 64;
static __device__ __align__(256) uint8_t scratch[kScratchBytes];
std::atomic<uint8_t*> g_scratch_ptr[kMaxDevices];      // per device, filled on the first call on a stream of that device
                                                        // (two racing first calls both look up the same address)

// The scratch of the device that owns `st` (every GPU holds its own copy of the array at its own address).  The first call
// per device looks the address up with that device current (restored afterwards); that look-up may load the code object, so it
// belongs outside a graph capture -- later calls only read the cached pointer.
int scratch_for_stream(const char* fn, hipStream_t st, uint8_t** out)
{
    int dev = -1;
    if (hipStreamGetDevice(st, &dev) != hipSuccess) {             // (the null stream: the current device)
        (void)hipGetLastError();
        IPX_HIP(hipGetDevice(&dev));
    }
    if (dev < 0 || dev >= kMaxDevices) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: device index %d outside 0..%d", fn, dev, kMaxDevices - 1);
    uint8_t* ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!ptr) {
        int cur = -1;
        IPX_HIP(hipGetDevice(&cur));
        if (cur != dev) IPX_HIP(hipSetDevice(dev));
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (cur != dev) (void)hipSetDevice(cur);
        if (e != hipSuccess || !p)
            return ipx_fail(IPX_ERR_HIP, "%s: hipGetSymbolAddress(scratch) failed on device %d: %s", fn, dev, hipGetErrorString(e));
        ptr = (uint8_t*)p;
        g_scratch_ptr[dev].store(ptr, std::memory_order_release);
    }
    *out = ptr;
    return IPX_OK;
}

// mode=log's table from an image's histogram -- server.py:385-388: s = log1p(c r) (float32, the 256 values come from the
// host, numpy's own log1p), s /= (s.max() + 1e-12), out = (s * 255.0).astype(uint8).  s increases with the level, so s.max()
// over the image is s at the highest level present; the 1e-12 is added in float32 (it only survives when that maximum is 0).
// One block of 256 threads per image; two separately rounded float32 operations per level, then truncation.  Levels above the
// highest one present are never looked up (numpy's cast of their out-of-range values is unspecified): they read 255.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void log_lut_kernel(const uint32_t* __restrict__ hist, const float* __restrict__ curve,
                                                      uint8_t* __restrict__ lut, int64_t lut_stride)
{
    __shared__ int s_top[4];
    const int t = (int)threadIdx.x;
    const uint32_t* h = hist + (size_t)blockIdx.x * 256;
    const unsigned long long present = __ballot(h[t] != 0u);           // levels 64 w .. 64 w + 63 of wave w
    if ((t & 63) == 0) s_top[t >> 6] = present ? 64 * (t >> 6) + 63 - __builtin_clzll(present) : -1;
    __syncthreads();
    int top = s_top[3] >= 0 ? s_top[3] : s_top[2] >= 0 ? s_top[2] : s_top[1] >= 0 ? s_top[1] : s_top[0];
    if (top < 0) top = 0;                                              // an all-zero histogram (no pixels): treat as level 0
    const float d = curve[top] + 1e-12f;
    const float v = __fdiv_rn(curve[t], d) * 255.0f;
    lut[(int64_t)blockIdx.x * lut_stride + t] = t > top || !(v < 256.0f) ? (uint8_t)255 : (uint8_t)(int)v;
}

}  // namespace

extern "C" {

size_t ipx_hist_workspace_bytes(int n, int h, int w)
{
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    return workspace_layout(n, h, w, nullptr, nullptr);
}

int ipx_hist_workspace_init(void* stream, void* workspace, size_t workspace_bytes)
{
    ipx_clear_error();
    if (!workspace) return ipx_fail(IPX_ERR_INVALID, "ipx_hist_workspace_init: null workspace");
    if ((uintptr_t)workspace & 3u) return ipx_fail(IPX_ERR_INVALID, "ipx_hist_workspace_init: workspace %p is not 4-byte aligned", workspace);
    if (workspace_bytes < kTicketBytes)
        return ipx_fail(IPX_ERR_INVALID, "ipx_hist_workspace_init: workspace %zu < %zu bytes (the smallest ipx_hist_workspace_bytes)",
                        workspace_bytes, kTicketBytes);
    IPX_HIP(hipMemsetAsync(workspace, 0, kTicketBytes, (hipStream_t)stream));
    return IPX_OK;
}

int ipx_hist_u8(void* stream, const uint8_t* src, int n, int h, int w, int64_t src_row_stride, int64_t src_img_stride,
                uint32_t* hist_dev, void* workspace, size_t workspace_bytes)
{
    if (int rc = check_hist_args("ipx_hist_u8", src, n, h, w, src_row_stride, src_img_stride)) return rc;
    if (!hist_dev) return ipx_fail(IPX_ERR_INVALID, "ipx_hist_u8: null histogram output");
    hipStream_t st = (hipStream_t)stream;
    void* own = nullptr;
    if (int rc = take_workspace("ipx_hist_u8", st, workspace_layout(n, h, w, nullptr, nullptr), &workspace, workspace_bytes, &own)) return rc;
    Workspace ws;
    workspace_layout(n, h, w, &ws, (uint8_t*)workspace);
    int rc = launch_hist(st, src, n, h, w, src_row_stride, src_img_stride, ws, hist_dev, nullptr);
    return release_workspace(st, own, rc);
}

int ipx_equalize_lut_u8(void* stream, const uint32_t* hist_dev, int n, uint8_t* lut_dev)
{
    ipx_clear_error();
    if (!hist_dev || !lut_dev) return ipx_fail(IPX_ERR_INVALID, "ipx_equalize_lut_u8: null pointer");
    if (n <= 0) return ipx_fail(IPX_ERR_INVALID, "ipx_equalize_lut_u8: n=%d must be positive", n);
    hipLaunchKernelGGL(hist_finish_kernel, dim3((unsigned)n), dim3(256 * kFinishGroups), 0, (hipStream_t)stream,
                       (const uint32_t*)nullptr, 0u, hist_dev, (uint32_t*)nullptr, lut_dev);
    IPX_LAUNCH_CHECK("hist_finish_kernel");
    return IPX_OK;
}

int ipx_log_lut_u8(void* stream, const uint32_t* hist_dev, int n, const float* curve_dev, uint8_t* lut_dev, int64_t lut_stride)
{
    ipx_clear_error();
    if (!hist_dev || !curve_dev || !lut_dev) return ipx_fail(IPX_ERR_INVALID, "ipx_log_lut_u8: null pointer");
    if (n <= 0) return ipx_fail(IPX_ERR_INVALID, "ipx_log_lut_u8: n=%d must be positive", n);
    if (lut_stride < 256) return ipx_fail(IPX_ERR_INVALID, "ipx_log_lut_u8: lut_stride %lld must be >= 256", (long long)lut_stride);
    hipLaunchKernelGGL(log_lut_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, hist_dev, curve_dev, lut_dev, lut_stride);
    IPX_LAUNCH_CHECK("log_lut_kernel");
    return IPX_OK;
}

int ipx_equalize_hist_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                         int64_t src_row_stride, int64_t src_img_stride, int64_t dst_row_stride, int64_t dst_img_stride,
                         uint32_t* hist_out_dev, uint8_t* lut_out_dev, void* workspace, size_t workspace_bytes)
{
    if (!dst) {                                             // tables only (the table pass is the caller's, e.g. ipx_sobel_lut_u8)
        ipx_clear_error();
        if (!hist_out_dev && !lut_out_dev) return ipx_fail(IPX_ERR_INVALID, "ipx_equalize_hist_u8: dst = NULL needs hist_out_dev or lut_out_dev");
        if (!src) return ipx_fail(IPX_ERR_INVALID, "ipx_equalize_hist_u8: null image pointer");
    } else if (int rc = ipx_check_image_args("ipx_equalize_hist_u8", src, dst, n, h, w, 1, 1, src_row_stride, src_img_stride,
                                             dst_row_stride, dst_img_stride)) return rc;
    if (int rc = check_hist_args("ipx_equalize_hist_u8", src, n, h, w, src_row_stride, src_img_stride)) return rc;
    hipStream_t st = (hipStream_t)stream;
    void* own = nullptr;
    if (int rc = take_workspace("ipx_equalize_hist_u8", st, workspace_layout(n, h, w, nullptr, nullptr), &workspace, workspace_bytes, &own))
        return rc;
    int rc = equalize_with_workspace(st, src, dst, n, h, w, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride,
                                     hist_out_dev, lut_out_dev, (uint8_t*)workspace);
    return release_workspace(st, own, rc);
}

int ipx_equalize_hist_noalloc_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                                 int64_t src_row_stride, int64_t src_img_stride, int64_t dst_row_stride, int64_t dst_img_stride,
                                 uint32_t* hist_out_dev, uint8_t* lut_out_dev)
{
    const char* fn = "ipx_equalize_hist_noalloc_u8";
    if (int rc = ipx_check_image_args(fn, src, dst, n, h, w, 1, 1, src_row_stride, src_img_stride,
                                      dst_row_stride, dst_img_stride)) return rc;
    if (int rc = check_hist_args(fn, src, n, h, w, src_row_stride, src_img_stride)) return rc;
    const size_t need = workspace_layout(n, h, w, nullptr, nullptr);
    if (need > kScratchBytes)
        return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: a batch of %d images needs %zu bytes of scratch, the built-in array holds %zu "
                        "(use ipx_equalize_hist_u8 with a workspace)", fn, n, need, kScratchBytes);
    uint8_t* ws = nullptr;
    if (int rc = scratch_for_stream(fn, (hipStream_t)stream, &ws)) return rc;
    // The only words of the scratch that have to start at zero are the tickets at its head: the array is zero when the code
    // object is loaded, and the block that wins a ticket puts it back to zero, so no call has to clear anything.  Every
    // partial count, histogram bin and table entry that is read is written first by the same call's own kernels.
    return equalize_with_workspace((hipStream_t)stream, src, dst, n, h, w, src_row_stride, src_img_stride,
                                   dst_row_stride, dst_img_stride, hist_out_dev, lut_out_dev, ws);
}

int ipx_bgr2gray_equalize_u8(void* stream, const uint8_t* src, uint8_t* gray, uint8_t* dst, int n, int h, int w, int src_channels,
                             int64_t src_row_stride, int64_t src_img_stride, int64_t gray_row_stride, int64_t gray_img_stride,
                             int64_t dst_row_stride, int64_t dst_img_stride,
                             uint32_t* hist_out_dev, uint8_t* lut_out_dev, void* workspace, size_t workspace_bytes)
{
    const char* fn = "ipx_bgr2gray_equalize_u8";
    if (src_channels != 3 && src_channels != 4) {
        ipx_clear_error();
        return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: src_channels=%d (3 = BGR or 4 = BGRA expected)", fn, src_channels);
    }
    if (int rc = ipx_check_image_args(fn, src, gray, n, h, w, src_channels, 1, src_row_stride, src_img_stride,
                                      gray_row_stride, gray_img_stride)) return rc;
    // dst == NULL: gray image, histograms and tables only (the caller applies the tables itself, e.g. together with
    // others in ipx_lut_multi_u8); at least one of hist_out_dev / lut_out_dev must then take the result
    if (dst) {
        if (int rc = ipx_check_image_args(fn, gray, dst, n, h, w, 1, 1, gray_row_stride, gray_img_stride,
                                          dst_row_stride, dst_img_stride)) return rc;
    } else if (!hist_out_dev && !lut_out_dev) {
        ipx_clear_error();
        return ipx_fail(IPX_ERR_INVALID, "%s: dst is null and neither hist_out_dev nor lut_out_dev is given", fn);
    }
    if (int rc = check_hist_args(fn, gray, n, h, w, gray_row_stride, gray_img_stride)) return rc;
    hipStream_t st = (hipStream_t)stream;
    void* own = nullptr;
    if (int rc = take_workspace(fn, st, workspace_layout(n, h, w, nullptr, nullptr), &workspace, workspace_bytes, &own)) return rc;
    Workspace ws;
    workspace_layout(n, h, w, &ws, (uint8_t*)workspace);
    uint8_t* lut = lut_out_dev ? lut_out_dev : ws.lut;
    GrayHistGeom g;
    g.rows = h; g.len = w; g.s_row = src_row_stride; g.s_img = src_img_stride; g.g_row = gray_row_stride; g.g_img = gray_img_stride;
    if (src_row_stride == (int64_t)w * src_channels && gray_row_stride == w) { g.rows = 1; g.len = (int64_t)w * h; }
    g.units_per_row = ipx_div_up((uint64_t)g.len, kGUnit);
    g.units_per_img = (unsigned)g.rows * g.units_per_row;
    // 56 KiB of LDS per block (histogram + transposition): 2 resident blocks per CU; never more blocks than the
    // workspace layout (sized by the default) holds
    unsigned bpi = blocks_per_image(n, h, w, 512);
    if (bpi > blocks_per_image(n, h, w)) bpi = blocks_per_image(n, h, w);
    if (bpi > g.units_per_img) bpi = g.units_per_img;
    const HistTail t = make_tail(ws, n, bpi, hist_out_dev, lut);
    const bool nt = ipx_use_nt((uint64_t)n * h * w * src_channels, kNtGrayHist);
    auto k = src_channels == 3 ? (nt ? gray_hist_kernel<3, true> : gray_hist_kernel<3, false>)
                               : (nt ? gray_hist_kernel<4, true> : gray_hist_kernel<4, false>);
    hipLaunchKernelGGL(k, dim3(bpi, (unsigned)n), dim3(kThreads), 0, st, src, gray, g, t);
    int rc = hipGetLastError() == hipSuccess ? IPX_OK : ipx_fail(IPX_ERR_HIP, "%s: launch of gray_hist_kernel failed", fn);
    if (rc == IPX_OK) rc = finish_if_needed(st, t, n, bpi);
    if (rc == IPX_OK && dst)
        rc = ipx_lut_dev_u8(stream, gray, dst, n, h, w, 1, gray_row_stride, gray_img_stride, dst_row_stride, dst_img_stride, lut, 256);
    return release_workspace(st, own, rc);
}

}  // extern "C"

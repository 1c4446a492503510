// ipx_hist.hip -- 256-bin histogram, equalizeHist LUT construction, cv2.equalizeHist.
//
// hist_partial_kernel: each block counts a slice of one image into an LDS histogram laid out
//   [256 bins][32 copies] (32 KiB): lane l always adds to copy (l & 31), i.e. to LDS bank (l & 31), so a
//   wave's 64 ds_add_u32 never collide on a bank whatever the pixel values are -- a constant image
//   (all lanes, one bin) costs the same as white noise.  The 32 copies are folded with a rotated
//   (conflict-free) read and the block stores its 256 partial counts with plain coalesced stores: no
//   global atomics, no zero-fill, bitwise reproducible.
// hist_finish_kernel: one block per image sums the per-block partials and (optionally) builds the
//   equalization table with an LDS scan, in float32 exactly as OpenCV does.
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

template <bool NT>
__global__ __launch_bounds__(kThreads) void hist_partial_kernel(const uint8_t* __restrict__ src, HistGeom g,
                                                                uint32_t* __restrict__ partial)
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
    // fold the 32 copies of bin `tid`; lane reads copy (j + tid) % 32 -> bank (j + tid) % 32: conflict-free
    if (tid < 256) {
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < kCopies; ++j) sum += s_hist[(tid << 5) | ((j + tid) & 31u)];
        partial[((size_t)img * gridDim.x + blockIdx.x) * 256 + tid] = sum;
    }
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
                                                             GrayHistGeom g, uint32_t* __restrict__ partial)
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
    if (tid < 256) {
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < kCopies; ++j) sum += s_hist[(tid << 5) | ((j + tid) & 31u)];
        partial[((size_t)img * gridDim.x + blockIdx.x) * 256 + tid] = sum;
    }
}

// One block per image: hist[bin] = sum over that image's partial blocks; optional equalization LUT.
// OpenCV histogram.cpp equalizeHist(): i0 = first non-zero bin; hist[i0]==total -> constant image maps to i0;
// scale = 255.f/(total-hist[i0]); lut[i0]=0; lut[i>i0] = saturate_cast<uchar>(cvRound(float(sum_i)*scale)).
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
    if (hist_out) hist_out[(size_t)img * 256 + tid] = h;
    if (!lut_out) return;

    if (tid == 0) s_i0 = 256;
    s_scan[0][tid] = h;
    __syncthreads();                                         // (exited waves no longer count at the barrier)
    if (h != 0) atomicMin(&s_i0, (int)tid);
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
    const int i0 = s_i0;
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

HistGeom make_geom(int h, int w, int64_t s_row, int64_t s_img)
{
    HistGeom g;
    g.rows = h; g.len = w; g.s_row = s_row; g.s_img = s_img;
    if (s_row == w) { g.rows = 1; g.len = (int64_t)w * h; }
    g.units_per_row = ipx_div_up((uint64_t)g.len, kUnit);
    g.units_per_img = (unsigned)g.rows * g.units_per_row;
    return g;
}

// blocks per image: enough blocks to fill the chip (LDS admits 4-5 blocks per CU), but at least
// 64 KiB of pixels per block so that zeroing/folding the 32 KiB LDS histogram stays in the noise.
unsigned blocks_per_image(int n, int h, int w, unsigned slots = 1024)
{
    const uint64_t bytes = (uint64_t)h * (uint64_t)w;
    uint64_t by_work = (bytes + 65535) / 65536;
#ifdef IPX_DEV
    static const int env_slots = [] { const char* e = getenv("IPX_HIST_SLOTS"); return e ? atoi(e) : 0; }();   // read once
    if (env_slots > 0) slots = (unsigned)env_slots;
#endif
    uint64_t by_chip = slots / (uint64_t)n;                      // 256 CUs x resident blocks per CU: ONE round (rounding up
                                                                 // put e.g. 24 images x 22 blocks = 528 blocks on 512 slots)
    uint64_t b = by_work < by_chip ? by_work : by_chip;
    return (unsigned)(b < 1 ? 1 : b);
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Workspace {
    uint32_t* partial; uint32_t* hist; uint8_t* lut;
};

size_t workspace_layout(int n, int h, int w, Workspace* ws, uint8_t* base)
{
    const size_t p = align256((size_t)n * blocks_per_image(n, h, w) * 256 * sizeof(uint32_t));
    const size_t hs = align256((size_t)n * 256 * sizeof(uint32_t));
    const size_t ls = align256((size_t)n * 256);
    if (ws) { ws->partial = (uint32_t*)base; ws->hist = (uint32_t*)(base + p); ws->lut = base + p + hs; }
    return p + hs + ls;
}

int check_hist_args(const char* fn, const uint8_t* src, int n, int h, int w, int64_t s_row, int64_t s_img)
{
    if (int rc = ipx_check_image_args(fn, src, src, n, h, w, 1, 1, s_row, s_img, s_row, s_img)) return rc;
    if ((uint64_t)h * (uint64_t)w > 0x7fffffffull)
        return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: %d x %d pixels exceed int32 counts (OpenCV's own limit)", fn, h, w);
    if (n > 65535) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: n=%d > 65535 images per call", fn, n);
    return IPX_OK;
}

int launch_partial(hipStream_t st, const uint8_t* src, int n, int h, int w, int64_t s_row, int64_t s_img,
                   uint32_t* partial, unsigned* nparts)
{
    HistGeom g = make_geom(h, w, s_row, s_img);
    unsigned bpi = blocks_per_image(n, h, w);
    if (bpi > g.units_per_img) bpi = g.units_per_img;     // strided rows: fewer units than bytes/64K suggests
    *nparts = bpi;
    // the histogram pass reads what the table pass re-reads right after: keep small batches in the Infinity Cache
    if (ipx_use_nt(2ull * (uint64_t)n * h * w, kNtHistThenLut))
        hipLaunchKernelGGL(hist_partial_kernel<true>, dim3(bpi, (unsigned)n), dim3(kThreads), 0, st, src, g, partial);
    else
        hipLaunchKernelGGL(hist_partial_kernel<false>, dim3(bpi, (unsigned)n), dim3(kThreads), 0, st, src, g, partial);
    IPX_LAUNCH_CHECK("hist_partial_kernel");
    return IPX_OK;
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
    unsigned nparts = 0;
    if (int rc = launch_partial(st, src, n, h, w, s_row, s_img, ws.partial, &nparts)) return rc;
    hipLaunchKernelGGL(hist_finish_kernel, dim3((unsigned)n), dim3(256 * kFinishGroups), 0, st, ws.partial, nparts,
                       (const uint32_t*)nullptr, hist_out_dev, lut);
    IPX_LAUNCH_CHECK("hist_finish_kernel");
    // the table pass (default cache policy up to kNtHistThenLut bytes: what the histogram pass read is still in the Infinity Cache)
    return ipx_lut_dev_u8((void*)st, src, dst, n, h, w, 1, s_row, s_img, d_row, d_img, lut, 256);
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

}  // namespace

extern "C" {

size_t ipx_hist_workspace_bytes(int n, int h, int w)
{
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    return workspace_layout(n, h, w, nullptr, nullptr);
}

int ipx_hist_u8(void* stream, const uint8_t* src, int n, int h, int w, int64_t src_row_stride, int64_t src_img_stride,
                uint32_t* hist_dev, void* workspace, size_t workspace_bytes)
{
    if (int rc = check_hist_args("ipx_hist_u8", src, n, h, w, src_row_stride, src_img_stride)) return rc;
    if (!hist_dev) return ipx_fail(IPX_ERR_INVALID, "ipx_hist_u8: null histogram output");
    hipStream_t st = (hipStream_t)stream;
    const size_t need = workspace_layout(n, h, w, nullptr, nullptr);
    void* own = nullptr;
    if (!workspace) {
        IPX_HIP(hipMallocAsync(&own, need, st));
        workspace = own;
    } else if (workspace_bytes < need) {
        return ipx_fail(IPX_ERR_INVALID, "ipx_hist_u8: workspace %zu < %zu bytes needed", workspace_bytes, need);
    }
    Workspace ws;
    workspace_layout(n, h, w, &ws, (uint8_t*)workspace);
    unsigned nparts = 0;
    int rc = launch_partial(st, src, n, h, w, src_row_stride, src_img_stride, ws.partial, &nparts);
    if (rc == IPX_OK) {
        hipLaunchKernelGGL(hist_finish_kernel, dim3((unsigned)n), dim3(256 * kFinishGroups), 0, st, ws.partial, nparts,
                           (const uint32_t*)nullptr, hist_dev, (uint8_t*)nullptr);
        if (hipGetLastError() != hipSuccess) rc = ipx_fail(IPX_ERR_HIP, "launch of hist_finish_kernel failed");
    }
    if (own) {
        hipError_t e = hipFreeAsync(own, st);
        if (e != hipSuccess && rc == IPX_OK) rc = ipx_fail(IPX_ERR_HIP, "hipFreeAsync failed: %s", hipGetErrorString(e));
    }
    return rc;
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

int ipx_equalize_hist_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                         int64_t src_row_stride, int64_t src_img_stride, int64_t dst_row_stride, int64_t dst_img_stride,
                         uint32_t* hist_out_dev, uint8_t* lut_out_dev, void* workspace, size_t workspace_bytes)
{
    if (int rc = ipx_check_image_args("ipx_equalize_hist_u8", src, dst, n, h, w, 1, 1, src_row_stride, src_img_stride,
                                      dst_row_stride, dst_img_stride)) return rc;
    if (int rc = check_hist_args("ipx_equalize_hist_u8", src, n, h, w, src_row_stride, src_img_stride)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t need = workspace_layout(n, h, w, nullptr, nullptr);
    void* own = nullptr;
    if (!workspace) {
        IPX_HIP(hipMallocAsync(&own, need, st));
        workspace = own;
    } else if (workspace_bytes < need) {
        return ipx_fail(IPX_ERR_INVALID, "ipx_equalize_hist_u8: workspace %zu < %zu bytes needed", workspace_bytes, need);
    }
    int rc = equalize_with_workspace(st, src, dst, n, h, w, src_row_stride, src_img_stride, dst_row_stride, dst_img_stride,
                                     hist_out_dev, lut_out_dev, (uint8_t*)workspace);
    if (own) {
        hipError_t e = hipFreeAsync(own, st);
        if (e != hipSuccess && rc == IPX_OK) rc = ipx_fail(IPX_ERR_HIP, "hipFreeAsync failed: %s", hipGetErrorString(e));
    }
    return rc;
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
    // Nothing in the scratch has to start at zero: every partial count, histogram bin and table entry that is read
    // is written first by this call's own kernels, so the previous call's contents are simply overwritten.
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
    const size_t need = workspace_layout(n, h, w, nullptr, nullptr);
    void* own = nullptr;
    if (!workspace) {
        IPX_HIP(hipMallocAsync(&own, need, st));
        workspace = own;
    } else if (workspace_bytes < need) {
        return ipx_fail(IPX_ERR_INVALID, "%s: workspace %zu < %zu bytes needed", fn, workspace_bytes, need);
    }
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
    const bool nt = ipx_use_nt((uint64_t)n * h * w * src_channels, kNtGrayHist);
    auto k = src_channels == 3 ? (nt ? gray_hist_kernel<3, true> : gray_hist_kernel<3, false>)
                               : (nt ? gray_hist_kernel<4, true> : gray_hist_kernel<4, false>);
    hipLaunchKernelGGL(k, dim3(bpi, (unsigned)n), dim3(kThreads), 0, st, src, gray, g, ws.partial);
    int rc = hipGetLastError() == hipSuccess ? IPX_OK : ipx_fail(IPX_ERR_HIP, "%s: launch of gray_hist_kernel failed", fn);
    if (rc == IPX_OK) {
        hipLaunchKernelGGL(hist_finish_kernel, dim3((unsigned)n), dim3(256 * kFinishGroups), 0, st, ws.partial, bpi,
                           (const uint32_t*)nullptr, hist_out_dev, lut);
        if (hipGetLastError() != hipSuccess) rc = ipx_fail(IPX_ERR_HIP, "%s: launch of hist_finish_kernel failed", fn);
    }
    if (rc == IPX_OK && dst)
        rc = ipx_lut_dev_u8(stream, gray, dst, n, h, w, 1, gray_row_stride, gray_img_stride, dst_row_stride, dst_img_stride, lut, 256);
    if (own) {
        hipError_t e = hipFreeAsync(own, st);
        if (e != hipSuccess && rc == IPX_OK) rc = ipx_fail(IPX_ERR_HIP, "hipFreeAsync failed: %s", hipGetErrorString(e));
    }
    return rc;
}

}  // extern "C"

// ipx_png.hip -- the un-filtering stage of PNG decoding on the GPU (decode half).
//
// cv2.imdecode (server.py:499, :252) inflates a PNG's IDAT stream and then undoes the per-row prediction filters
// (None / Sub / Up / Average / Paeth, PNG specification section 9: every byte is stored minus a prediction from its left,
// upper and upper-left neighbours, modulo 256).  The host inflates (zlib, off the GIL, on the codec pool) and uploads the
// filtered scan lines as they are -- the same number of bytes as the pixels -- and this kernel reconstructs the pixels.
//
// The recurrence runs along a row (left neighbour) AND down the image (upper neighbours), so the only parallelism is a
// wavefront: a lane per row, lane l one pixel behind lane l - 1.  A wave owns 64 consecutive rows; at every step a lane
// takes its upper neighbour's pixel from the lane above with one DPP shift (wave_shr:1) and keeps the previous one as the
// upper-left.  Row 63 of a wave is the upper row of the next wave's row 0: lane 63 writes its pixels to an edge buffer with
// write-through stores and, every 32 pixels, publishes how far it has got (device-scope store after its stores were
// acknowledged); lane 0 of the next block polls that counter before it enters a new 32-pixel chunk.  Blocks are dispatched
// in index order, so the block a wave waits for is always resident or finished; the poll gives up after a bounded number
// of tries and reports it (no wave can spin for ever).  One 4K RGB frame: 34 waves, ~7000 steps.
//
// Depth-agnostic: the filters work on bytes, `bpp` = bytes per complete pixel (1, 2, 3, 4, 6, 8), as in the specification;
// 16-bit samples come out big-endian as they are in the file.
#include "ipx_common.h"

namespace {

struct PngGeom {
    int h, w;                                   // rows, pixels per row
    int64_t raw_stride, d_row;                  // bytes between filtered rows (1 + w * bpp, or more), between output rows
    unsigned nblocks;
};

constexpr int kPf = 8;                          // pixels loaded ahead of their use
constexpr int kPublish = 32;                    // pixels between progress updates of a wave's last row
constexpr unsigned kMaxPolls = 1u << 22;

template <int NW> struct Unit { uint32_t w[NW]; };

__device__ __forceinline__ uint32_t ld_dev(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int BPP>
__global__ __launch_bounds__(64) void png_unfilter_kernel(const uint8_t* __restrict__ raw, uint8_t* __restrict__ dst, PngGeom g,
                                                          uint32_t* __restrict__ edge, uint32_t* __restrict__ progress,
                                                          uint32_t* __restrict__ status)
{
    constexpr int NW = (BPP + 3) / 4;
    typedef Unit<NW> U;
    const int lane = (int)threadIdx.x;
    const unsigned blk = blockIdx.x;
    const int y = (int)blk * 64 + lane;
    const bool valid = y < g.h;
    const uint8_t* frow = raw + (int64_t)(valid ? y : g.h - 1) * g.raw_stride;
    uint8_t* orow = dst + (int64_t)(valid ? y : g.h - 1) * g.d_row;
    int t = valid ? (int)frow[0] : 0;
    if (t > 4) {                                                      // not a PNG filter type: report, reconstruct as None
        atomicOr(status, 1u);
        t = 0;
    }
    frow += 1;
    const uint32_t* edge_in = edge + (size_t)(blk ? blk - 1 : 0) * (size_t)g.w * NW;   // the row above this wave's row 0
    uint32_t* edge_out = edge + (size_t)blk * (size_t)g.w * NW;
    const bool has_above = blk > 0, feeds_next = blk + 1 < g.nblocks;

    auto load_unit = [&](int p) -> U {
        U u;
#pragma unroll
        for (int i = 0; i < NW; ++i) u.w[i] = 0u;
        if (valid && p >= 0 && p < g.w) {
            const uint8_t* s = frow + (int64_t)p * BPP;
#pragma unroll
            for (int i = 0; i < BPP; ++i) u.w[i >> 2] |= (uint32_t)s[i] << (8 * (i & 3));
        }
        return u;
    };
    auto load_edge = [&](int p) -> U {
        U u;
#pragma unroll
        for (int i = 0; i < NW; ++i) u.w[i] = 0u;
        if (has_above && lane == 0 && p < g.w) {
#pragma unroll
            for (int i = 0; i < NW; ++i) u.w[i] = ld_dev(edge_in + (size_t)p * NW + i);
        }
        return u;
    };
    // the row above must have published `need` pixels (wave-uniform: every lane polls the same word)
    auto wait_for = [&](int need) {
        if (!has_above) return;
        need = need < g.w ? need : g.w;
        unsigned polls = 0;
        while ((int)ld_dev(progress + (blk - 1)) < need) {
            __builtin_amdgcn_s_sleep(2);
            if (++polls > kMaxPolls) {
                if (lane == 0) atomicOr(status, 2u);
                break;
            }
        }
    };

    U fq[kPf], eq[kPf];
    wait_for(kPublish + kPf);
#pragma unroll
    for (int i = 0; i < kPf; ++i) {
        fq[i] = load_unit(i - lane);
        eq[i] = load_edge(i);
    }
    U cur, bprev;
#pragma unroll
    for (int i = 0; i < NW; ++i) cur.w[i] = bprev.w[i] = 0u;

    const int nsteps = g.w + 64;
    for (int s0 = 0; s0 < nsteps; s0 += kPf) {
#pragma unroll
        for (int i = 0; i < kPf; ++i) {
            const int s = s0 + i, p = s - lane;
            if ((s & (kPublish - 1)) == 0 && s > 0 && s < g.w) wait_for(s + kPublish + kPf);
            const U f = fq[i], e = eq[i];
            fq[i] = load_unit(p + kPf);
            eq[i] = load_edge(s + kPf);
            // upper neighbour: the pixel the lane above produced one step ago (lane 0: the row above this wave, or nothing)
            U b;
#pragma unroll
            for (int k = 0; k < NW; ++k)
                b.w[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)e.w[k], (int)cur.w[k], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const bool first = p <= 0;                                // no left neighbours
            U out;
#pragma unroll
            for (int k = 0; k < NW; ++k) out.w[k] = 0u;
#pragma unroll
            for (int c = 0; c < BPP; ++c) {
                const int sh = 8 * (c & 3), k = c >> 2;
                const int fa = (int)((f.w[k] >> sh) & 255u), bb = (int)((b.w[k] >> sh) & 255u);
                const int aa = first ? 0 : (int)((cur.w[k] >> sh) & 255u), cc = first ? 0 : (int)((bprev.w[k] >> sh) & 255u);
                const int pa = __builtin_abs(bb - cc), pb = __builtin_abs(aa - cc), pc = __builtin_abs(aa + bb - 2 * cc);
                const int paeth = (pa <= pb && pa <= pc) ? aa : (pb <= pc ? bb : cc);
                const int pred = t == 1 ? aa : t == 2 ? bb : t == 3 ? (aa + bb) >> 1 : t == 4 ? paeth : 0;
                out.w[k] |= (uint32_t)((fa + pred) & 255) << sh;
            }
            const bool active = valid && p >= 0 && p < g.w;
            if (active) {
                uint8_t* o = orow + (int64_t)p * BPP;
#pragma unroll
                for (int c = 0; c < BPP; ++c) o[c] = (uint8_t)(out.w[c >> 2] >> (8 * (c & 3)));
                if (feeds_next && lane == 63) {
#pragma unroll
                    for (int k = 0; k < NW; ++k) st_dev(edge_out + (size_t)p * NW + k, out.w[k]);
                }
            }
            bprev = b;
            cur = out;
            // lane 63 has finished pixel s - 63: publish every kPublish pixels and at the row's end
            const int p63 = s - 63;
            if (feeds_next && p63 >= 0 && p63 < g.w && ((p63 & (kPublish - 1)) == kPublish - 1 || p63 == g.w - 1)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the edge stores are acknowledged (write-through)
                if (lane == 63) st_dev(progress + blk, (uint32_t)(p63 + 1));
            }
        }
    }
}

template <int BPP>
void launch_png(hipStream_t st, const uint8_t* raw, uint8_t* dst, const PngGeom& g, uint32_t* edge, uint32_t* progress, uint32_t* status)
{
    hipLaunchKernelGGL(png_unfilter_kernel<BPP>, dim3(g.nblocks), dim3(64), 0, st, raw, dst, g, edge, progress, status);
}

size_t png_layout(int h, int w, int bpp, size_t* progress_bytes)
{
    const size_t nblocks = ((size_t)h + 63) / 64;
    const size_t pb = (nblocks * sizeof(uint32_t) + 255) & ~(size_t)255;
    if (progress_bytes) *progress_bytes = pb;
    return pb + nblocks * (size_t)w * (size_t)((bpp + 3) / 4) * sizeof(uint32_t);
}

}  // namespace

extern "C" {

size_t ipx_png_unfilter_workspace_bytes(int h, int w, int bytes_per_pixel)
{
    if (h <= 0 || w <= 0 || bytes_per_pixel <= 0) return 0;
    return png_layout(h, w, bytes_per_pixel, nullptr);
}

int ipx_png_unfilter_u8(void* stream, const uint8_t* raw, int64_t raw_row_stride, uint8_t* dst, int64_t dst_row_stride,
                        int h, int w, int bytes_per_pixel, void* workspace, size_t workspace_bytes, uint32_t* status_dev)
{
    const char* fn = "ipx_png_unfilter_u8";
    ipx_clear_error();
    if (!raw || !dst || !workspace || !status_dev) return ipx_fail(IPX_ERR_INVALID, "%s: null pointer", fn);
    if (h <= 0 || w <= 0) return ipx_fail(IPX_ERR_INVALID, "%s: empty image %d x %d", fn, h, w);
    const int bpp = bytes_per_pixel;
    if (bpp != 1 && bpp != 2 && bpp != 3 && bpp != 4 && bpp != 6 && bpp != 8)
        return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: %d bytes per pixel (1, 2, 3, 4, 6 or 8: gray / gray+alpha / RGB / RGBA at 8 or 16 bits)", fn, bpp);
    if ((int64_t)w * bpp > (1 << 30)) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: row of %lld bytes too long", fn, (long long)w * bpp);
    if (raw_row_stride < (int64_t)w * bpp + 1) return ipx_fail(IPX_ERR_INVALID, "%s: raw row stride %lld < 1 + %lld", fn, (long long)raw_row_stride, (long long)w * bpp);
    if (dst_row_stride < (int64_t)w * bpp) return ipx_fail(IPX_ERR_INVALID, "%s: dst row stride %lld < %lld", fn, (long long)dst_row_stride, (long long)w * bpp);
    if ((uintptr_t)workspace & 3u) return ipx_fail(IPX_ERR_INVALID, "%s: workspace %p is not 4-byte aligned", fn, workspace);
    size_t pb = 0;
    const size_t need = png_layout(h, w, bpp, &pb);
    if (workspace_bytes < need) return ipx_fail(IPX_ERR_INVALID, "%s: workspace %zu < %zu bytes needed", fn, workspace_bytes, need);
    PngGeom g;
    g.h = h; g.w = w; g.raw_stride = raw_row_stride; g.d_row = dst_row_stride;
    g.nblocks = (unsigned)(((size_t)h + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
    uint32_t* progress = (uint32_t*)workspace;
    uint32_t* edge = (uint32_t*)((uint8_t*)workspace + pb);
    IPX_HIP(hipMemsetAsync(progress, 0, pb, st));
    IPX_HIP(hipMemsetAsync(status_dev, 0, sizeof(uint32_t), st));
    switch (bpp) {
    case 1: launch_png<1>(st, raw, dst, g, edge, progress, status_dev); break;
    case 2: launch_png<2>(st, raw, dst, g, edge, progress, status_dev); break;
    case 3: launch_png<3>(st, raw, dst, g, edge, progress, status_dev); break;
    case 4: launch_png<4>(st, raw, dst, g, edge, progress, status_dev); break;
    case 6: launch_png<6>(st, raw, dst, g, edge, progress, status_dev); break;
    default: launch_png<8>(st, raw, dst, g, edge, progress, status_dev); break;
    }
    IPX_LAUNCH_CHECK("png_unfilter_kernel");
    return IPX_OK;
}

}  // extern "C"

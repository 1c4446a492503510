// ipx_core.hip -- error plumbing, argument validation and host-side helpers of libipx.
#include "ipx_common.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

int ipx_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

void ipx_clear_error() { g_err[0] = 0; }

int ipx_check_disjoint(const char* fn, const char* what_a, const void* a, int64_t a_row_bytes, int64_t a_row, int64_t a_img,
                       const char* what_b, const void* b, int64_t b_row_bytes, int64_t b_row, int64_t b_img, int n, int h)
{
    if (!a || !b || n <= 0 || h <= 0) return IPX_OK;                 // (null pointers and sizes are reported by the argument check)
    const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
    const uintptr_t a1 = a0 + (uintptr_t)((n - 1) * a_img + (h - 1) * a_row + a_row_bytes);
    const uintptr_t b1 = b0 + (uintptr_t)((n - 1) * b_img + (h - 1) * b_row + b_row_bytes);
    if (a0 < b1 && b0 < a1) {
        // two column ranges of ONE pitched buffer (left and right half of a frame, say) interleave without sharing a byte
        if (a_row == b_row && a_row > 0 && (n == 1 || a_img == b_img) && (n == 1 || a_img % a_row == 0)) {
            const int64_t d = (int64_t)(b0 % (uintptr_t)a_row) - (int64_t)(a0 % (uintptr_t)a_row);
            const int64_t c = d >= 0 ? d : d + a_row;                    // b's first column relative to a's, in [0, pitch)
            if (c >= a_row_bytes && c + b_row_bytes <= a_row) return IPX_OK;
        }
        return ipx_fail(IPX_ERR_INVALID, "%s: %s [%p, +%llu) overlaps %s [%p, +%llu): the operator reads neighbouring pixels and cannot work in place",
                        fn, what_a, a, (unsigned long long)(a1 - a0), what_b, b, (unsigned long long)(b1 - b0));
    }
    return IPX_OK;
}

int ipx_check_image_args(const char* fn, const void* src, const void* dst, int n, int h, int w,
                         int src_c, int dst_c, int64_t src_row_stride, int64_t src_img_stride,
                         int64_t dst_row_stride, int64_t dst_img_stride)
{
    ipx_clear_error();
    if (!src || !dst) return ipx_fail(IPX_ERR_INVALID, "%s: null image pointer", fn);
    if (n <= 0 || h <= 0 || w <= 0) return ipx_fail(IPX_ERR_INVALID, "%s: n=%d h=%d w=%d must be positive", fn, n, h, w);
    if (src_c < 1 || src_c > 4 || dst_c < 1 || dst_c > 4)
        return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: channel count %d/%d outside 1..4", fn, src_c, dst_c);
    if ((int64_t)w * 4 > 0x7fffffff) return ipx_fail(IPX_ERR_UNSUPPORTED, "%s: w=%d too large", fn, w);
    if (src_row_stride < (int64_t)w * src_c) return ipx_fail(IPX_ERR_INVALID, "%s: src_row_stride %lld < %lld row bytes", fn, (long long)src_row_stride, (long long)w * src_c);
    if (dst_row_stride < (int64_t)w * dst_c) return ipx_fail(IPX_ERR_INVALID, "%s: dst_row_stride %lld < %lld row bytes", fn, (long long)dst_row_stride, (long long)w * dst_c);
    if (n > 1) {
        if (src_img_stride < src_row_stride * (int64_t)(h - 1) + (int64_t)w * src_c)
            return ipx_fail(IPX_ERR_INVALID, "%s: src_img_stride %lld smaller than one image", fn, (long long)src_img_stride);
        if (dst_img_stride < dst_row_stride * (int64_t)(h - 1) + (int64_t)w * dst_c)
            return ipx_fail(IPX_ERR_INVALID, "%s: dst_img_stride %lld smaller than one image", fn, (long long)dst_img_stride);
    }
    return IPX_OK;
}

bool ipx_use_nt(uint64_t bytes_touched, uint64_t site_threshold)
{
    static const long long env_thr = [] {
        const char* v = getenv("IPX_NT_BYTES");
        return v && *v ? atoll(v) : -1ll;
    }();
    return bytes_touched >= (env_thr >= 0 ? (uint64_t)env_thr : site_threshold);
}

extern "C" {

int ipx_version(void) { return 100; /* 0.1.0 */ }

const char* ipx_last_error(void) { return g_err; }

int ipx_check_device(void)
{
    ipx_clear_error();
    int dev = 0;
    IPX_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    IPX_HIP(hipGetDeviceProperties(&p, dev));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return ipx_fail(IPX_ERR_UNSUPPORTED, "device %d is %s; libipx is built for gfx950 only", dev, p.gcnArchName);
    return IPX_OK;
}

// Q0.8 Gaussian taps as cv2.GaussianBlur builds them for uint8 input (server.py:428):
// OpenCV smooth.dispatch.cpp getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED.
int ipx_gauss_kernel_q8(int k, double sigma, uint16_t* out)
{
    ipx_clear_error();
    if (!out) return ipx_fail(IPX_ERR_INVALID, "ipx_gauss_kernel_q8: null output");
    if (k < 1 || (k & 1) == 0 || k > IPX_MAX_GAUSS_K)
        return ipx_fail(IPX_ERR_INVALID, "ipx_gauss_kernel_q8: k=%d must be odd and in 1..%d", k, IPX_MAX_GAUSS_K);
    double f[IPX_MAX_GAUSS_K];                  // (no heap: nothing here may throw across the C ABI)
    static const double t3[] = {0.25, 0.5, 0.25};
    static const double t5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
    static const double t7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
    static const double t9[] = {4 / 256., 13 / 256., 30 / 256., 51 / 256., 60 / 256., 51 / 256., 30 / 256., 13 / 256., 4 / 256.};
    const double* tab = nullptr;
    if (sigma <= 0) tab = k == 3 ? t3 : k == 5 ? t5 : k == 7 ? t7 : k == 9 ? t9 : nullptr;
    if (k == 1 && sigma <= 0) { out[0] = 256; return IPX_OK; }
    if (tab) {
        for (int i = 0; i < k; ++i) f[i] = tab[i];
    } else {
        // getGaussianKernelBitExact's order of operations: sigma by ONE fused multiply-add, x = 2 (i - (k-1)/2) as an
        // integer with scale -0.125 / sigma^2, only the first half of the taps summed, sum = 2 * half + 1 (the centre tap
        // is exp(0)), every tap MULTIPLIED by 1 / sum.  (OpenCV evaluates exp in softfloat, this uses libm's: both are
        // within an ulp; tests/test_oracle_golden.py shows no tap of k <= 255 sits near enough to a rounding tie for
        // that to matter.)
        const double sg = sigma > 0 ? sigma : fma((double)k, 0.15, 0.35);
        const double scale2 = -0.125 / (sg * sg);
        double half = 0;
        for (int i = 0, x = 1 - k; i < k / 2; ++i, x += 2) { f[i] = exp((double)(x * x) * scale2); half += f[i]; }
        const double mul1 = 1.0 / (half * 2.0 + 1.0);
        for (int i = 0; i < k / 2; ++i) f[i] *= mul1;
        f[k / 2] = mul1;
    }
    double err = 0;
    long acc = 0;
    for (int i = 0; i < k / 2; ++i) {           // error diffusion from the ends inward
        const double adj = f[i] * 256.0 + err;
        const long v0 = lrint(adj);              // cvRound
        err = adj - (double)v0;
        out[i] = out[k - 1 - i] = (uint16_t)v0;
        acc += v0;
    }
    out[k / 2] = (uint16_t)(256 - 2 * acc);      // centre tap closes the sum to exactly 256
    return IPX_OK;
}

}  // extern "C"

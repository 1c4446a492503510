/* CPU oracle, plain C -- TEST INFRASTRUCTURE ONLY (checker + bench cpu_baseline "port").
 *
 * A scalar C restatement of the reference hot path (/root/reference/server.py), written
 * independently of oracle/oracle.py so that the two restatements check each other.  It is
 * never linked into, loaded by or called from the product library (libipx.so).
 *
 * OpenCV-backed steps follow the published OpenCV 4.x algorithms (opencv-python-headless
 * 4.12.0.88, reference requirements.txt:9) -- PARITY UNPINNED against a running OpenCV, see
 * oracle/oracle.py header.  box3/sobel have no reference implementation (build's own).
 *
 * Build: make -C oracle   ->  oracle/_build/liboracle.so   (gcc -O2 -fopenmp)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int reflect101(int p, int len)
{   /* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p; else p = 2 * (len - 1) - p;
    }
    return p;
}

/* server.py:516-517  cv2.cvtColor(BGR2GRAY), 8-bit: Q15 coefficients 3735/19235/9798 */
void orc_bgr2gray_u8(const uint8_t* src, uint8_t* dst, int h, int w, int cn)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = src + (size_t)y * w * cn;
        uint8_t* d = dst + (size_t)y * w;
        for (int x = 0; x < w; ++x, s += cn)
            d[x] = (uint8_t)((3735u * s[0] + 19235u * s[1] + 9798u * s[2] + 16384u) >> 15);
    }
}

void orc_lut_u8(const uint8_t* src, uint8_t* dst, size_t n, const uint8_t* lut)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) dst[i] = lut[src[i]];
}

void orc_hist_u8(const uint8_t* src, size_t n, uint32_t* hist)
{
    memset(hist, 0, 256 * sizeof(uint32_t));
#pragma omp parallel
    {
        uint32_t local[256] = {0};
#pragma omp for schedule(static) nowait
        for (size_t i = 0; i < n; ++i) local[src[i]]++;
#pragma omp critical
        for (int i = 0; i < 256; ++i) hist[i] += local[i];
    }
}

/* server.py:246  cv2.equalizeHist -> LUT construction (histogram.cpp) */
void orc_equalize_lut(const uint32_t* hist, uint8_t* lut)
{
    int64_t total = 0;
    for (int i = 0; i < 256; ++i) total += hist[i];
    memset(lut, 0, 256);
    if (total == 0) return;
    int i = 0;
    while (!hist[i]) ++i;
    if ((int64_t)hist[i] == total) { memset(lut, i, 256); return; }
    float scale = 255.f / (float)(int)(total - hist[i]);
    int sum = 0;
    for (lut[i++] = 0; i < 256; ++i) {
        sum += (int)hist[i];
        float v = (float)sum * scale;
        long r = lrintf(v);                  /* cvRound: round half to even */
        lut[i] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
    }
}

void orc_equalize_hist_u8(const uint8_t* src, uint8_t* dst, size_t n)
{
    uint32_t hist[256]; uint8_t lut[256];
    orc_hist_u8(src, n, hist);
    orc_equalize_lut(hist, lut);
    orc_lut_u8(src, dst, n, lut);
}

/* 3x3 family on interleaved cn-channel images, REFLECT_101.  op: 0 sharpen (server.py:263-277:
 * clip(5c-n-s-e-w)), 1 box (round(sum9/9)), 2 sobel (min(255,|gx|+|gy|)). */
void orc_conv3_u8(const uint8_t* src, uint8_t* dst, int h, int w, int cn, int op)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const uint8_t* r0 = src + (size_t)reflect101(y - 1, h) * w * cn;
        const uint8_t* r1 = src + (size_t)y * w * cn;
        const uint8_t* r2 = src + (size_t)reflect101(y + 1, h) * w * cn;
        uint8_t* d = dst + (size_t)y * w * cn;
        for (int x = 0; x < w; ++x) {
            int xl = reflect101(x - 1, w) * cn, xc = x * cn, xr = reflect101(x + 1, w) * cn;
            for (int c = 0; c < cn; ++c) {
                int a = r0[xl + c], b = r0[xc + c], cc = r0[xr + c];
                int dd = r1[xl + c], e = r1[xc + c], f = r1[xr + c];
                int g = r2[xl + c], hh = r2[xc + c], i = r2[xr + c];
                int v;
                if (op == 0) { v = 5 * e - (b + hh + dd + f); v = v < 0 ? 0 : v > 255 ? 255 : v; }
                else if (op == 1) { v = (a + b + cc + dd + e + f + g + hh + i + 4) / 9; }
                else {
                    int gx = (cc + 2 * f + i) - (a + 2 * dd + g);
                    int gy = (g + 2 * hh + i) - (a + 2 * b + cc);
                    v = abs(gx) + abs(gy); if (v > 255) v = 255;
                }
                d[xc + c] = (uint8_t)v;
            }
        }
    }
}

/* server.py:428  cv2.GaussianBlur 8-bit fixed-point path: kernel builders */
static const double small_tab[5][9] = {
    {1.0}, {0.25, 0.5, 0.25}, {0.0625, 0.25, 0.375, 0.25, 0.0625},
    {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125},
    {4 / 256., 13 / 256., 30 / 256., 51 / 256., 60 / 256., 51 / 256., 30 / 256., 13 / 256., 4 / 256.}};

int orc_gauss_kernel_q8(int k, double sigma, uint16_t* out)
{
    if (k < 1 || !(k & 1) || k > 1023) return -1;
    double* kf = (double*)malloc(sizeof(double) * k);
    if (sigma <= 0 && k <= 9) {
        for (int i = 0; i < k; ++i) kf[i] = small_tab[k / 2][i];
    } else {
        /* getGaussianKernelBitExact's order: sigma by one fused multiply-add, integer abscissa x = 2i-(k-1) with scale
           -0.125/sigma^2, first half summed, sum = 2*half + 1, taps multiplied by 1/sum */
        double sg = sigma > 0 ? sigma : fma((double)k, 0.15, 0.35), scale2 = -0.125 / (sg * sg), half = 0;
        for (int i = 0, x = 1 - k; i < k / 2; ++i, x += 2) { kf[i] = exp((double)(x * x) * scale2); half += kf[i]; }
        double mul1 = 1.0 / (half * 2.0 + 1.0);
        for (int i = 0; i < k / 2; ++i) { kf[i] *= mul1; kf[k - 1 - i] = kf[i]; }
        kf[k / 2] = mul1;
    }
    double err = 0; int acc = 0;
    for (int i = 0; i < k / 2; ++i) {
        double adj = kf[i] * 256.0 + err;
        int v0 = (int)lrint(adj);
        err = adj - v0;
        out[i] = out[k - 1 - i] = (uint16_t)v0; acc += v0;
    }
    out[k / 2] = (uint16_t)(256 - 2 * acc);
    free(kf);
    return 0;
}

/* separable Q0.8 x Q0.8 filter: row pass to Q8.8, column pass to Q8.16, (acc+2^15)>>16 */
void orc_sep_q8_u8(const uint8_t* src, uint8_t* dst, int h, int w, int cn,
                   const uint16_t* kx, int nkx, const uint16_t* ky, int nky)
{
    size_t L = (size_t)w * cn;
    uint16_t* tmp = (uint16_t*)malloc((size_t)h * L * sizeof(uint16_t));
    int rx = nkx / 2, ry = nky / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = src + (size_t)y * L;
        uint16_t* t = tmp + (size_t)y * L;
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < cn; ++c) {
                uint32_t acc = 0;
                for (int j = 0; j < nkx; ++j) acc += (uint32_t)kx[j] * s[(size_t)reflect101(x + j - rx, w) * cn + c];
                t[(size_t)x * cn + c] = (uint16_t)acc;
            }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        uint8_t* d = dst + (size_t)y * L;
        for (size_t b = 0; b < L; ++b) {
            uint32_t acc = 0;
            for (int i = 0; i < nky; ++i) acc += (uint32_t)ky[i] * tmp[(size_t)reflect101(y + i - ry, h) * L + b];
            acc = (acc + 32768u) >> 16;
            d[b] = (uint8_t)(acc > 255 ? 255 : acc);
        }
    }
    free(tmp);
}

/* PNG reconstruction (un-filtering), PNG specification (ISO/IEC 15948) section 9.2-9.4: what libpng does inside
 * cv2.imdecode (server.py:499, :252) after inflating IDAT.  raw: h rows of [type][w*bpp filtered bytes] `stride` apart;
 * out: h rows of w*bpp bytes.  Returns 0, or 1 + the first row whose type byte is above 4. */
int orc_png_unfilter(const uint8_t* raw, size_t stride, uint8_t* out, int h, int w, int bpp)
{
    const size_t L = (size_t)w * bpp;
    for (int y = 0; y < h; ++y) {
        const uint8_t* f = raw + (size_t)y * stride;
        const int t = f[0];
        f += 1;
        uint8_t* x = out + (size_t)y * L;
        const uint8_t* up = y ? x - L : NULL;
        if (t > 4) return 1 + y;
        for (size_t i = 0; i < L; ++i) {
            const int a = i >= (size_t)bpp ? x[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
            int pred = 0;
            if (t == 1) pred = a;
            else if (t == 2) pred = b;
            else if (t == 3) pred = (a + b) >> 1;
            else if (t == 4) {
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            x[i] = (uint8_t)(f[i] + pred);
        }
    }
    return 0;
}

/* the inverse, for building test streams: filter row y with type types[y] (forward filters of section 9.2) */
void orc_png_filter(const uint8_t* img, uint8_t* raw, size_t stride, int h, int w, int bpp, const uint8_t* types)
{
    const size_t L = (size_t)w * bpp;
    for (int y = 0; y < h; ++y) {
        const uint8_t* x = img + (size_t)y * L;
        const uint8_t* up = y ? x - L : NULL;
        uint8_t* f = raw + (size_t)y * stride;
        const int t = types[y];
        f[0] = (uint8_t)t;
        f += 1;
        for (size_t i = 0; i < L; ++i) {
            const int a = i >= (size_t)bpp ? x[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
            int pred = 0;
            if (t == 1) pred = a;
            else if (t == 2) pred = b;
            else if (t == 3) pred = (a + b) >> 1;
            else if (t == 4) {
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            f[i] = (uint8_t)(x[i] - pred);
        }
    }
}

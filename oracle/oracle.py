"""CPU oracle for the image hot path -- TEST INFRASTRUCTURE ONLY.

This module is a numpy restatement of the arithmetic that the reference
(`/root/reference/server.py`) performs on the `/api/process` hot path.  It is
the *checker*: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it.  Nothing under `image-processing_amd/` imports
it, and the product path fails loudly when the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):

* numpy-only branches (negative, gamma, log, piecewise, lee_filter) are pinned
  by `tests/golden/server_numpy_ops.npz`, which `oracle/make_golden.py` made by
  driving the real `server.py` through Flask's test client.
* OpenCV-backed branches (cvtColor BGR2GRAY, equalizeHist, threshold,
  Laplacian, GaussianBlur) follow the published OpenCV 4.x algorithms
  (`opencv-python-headless==4.12.0.88`, reference `requirements.txt:9`).  That
  wheel is absent from this image, the reference holds no fixtures for these
  calls, so for them the status is **parity unpinned**.
* 3x3 box blur and Sobel have no implementation in the reference at all
  (SURVEY.md section 8(a) A13); the definitions here are this build's own.

Every function cites the reference line it restates.
"""
from __future__ import annotations

import math

import numpy as np

# --------------------------------------------------------------------------
# A2  cv2.cvtColor(img, COLOR_BGR2GRAY) on uint8     server.py:516-517
# --------------------------------------------------------------------------
# OpenCV 4.x color_rgb.simd.hpp, RGB2Gray<uchar>: 15-bit fixed point,
# coefficients BY15=3735, GY15=19235, RY15=9798 (sum 32768), rounding add 1<<14.
GRAY_B, GRAY_G, GRAY_R, GRAY_SHIFT = 3735, 19235, 9798, 15


def bgr2gray_u8(bgr: np.ndarray) -> np.ndarray:
    """`H x W x {3,4}` BGR(A) uint8 -> `H x W` uint8.  A 4th channel is dropped
    first, as `cv2.cvtColor(img, COLOR_BGRA2BGR)` does (server.py:509-513)."""
    assert bgr.dtype == np.uint8 and bgr.ndim == 3 and bgr.shape[2] in (3, 4)
    b = bgr[..., 0].astype(np.uint32)
    g = bgr[..., 1].astype(np.uint32)
    r = bgr[..., 2].astype(np.uint32)
    y = (GRAY_B * b + GRAY_G * g + GRAY_R * r + (1 << (GRAY_SHIFT - 1))) >> GRAY_SHIFT
    return y.astype(np.uint8)


def bgr2gray_u16(bgr16: np.ndarray) -> np.ndarray:
    """`cv2.cvtColor(img, COLOR_BGR2GRAY)` on a uint16 image, server.py:516-517 when the file is a 16-bit colour PNG:
    OpenCV 4.x RGB2Gray<ushort> uses the same Q15 coefficients as the byte version (gray_shift = 15),
    `(ushort)CV_DESCALE((unsigned)(b*cb + g*cg + r*cr), 15)`.  A 4th channel is dropped first (server.py:509-513).
    Parity unpinned, as for every cv2 call (module docstring)."""
    assert bgr16.dtype == np.uint16 and bgr16.ndim == 3 and bgr16.shape[2] in (3, 4)
    b = bgr16[..., 0].astype(np.uint64)
    g = bgr16[..., 1].astype(np.uint64)
    r = bgr16[..., 2].astype(np.uint64)
    y = (GRAY_B * b + GRAY_G * g + GRAY_R * r + (1 << (GRAY_SHIFT - 1))) >> GRAY_SHIFT
    assert int(y.max(initial=0)) <= 65535
    return y.astype(np.uint16)


def gray_u8_from_u16(img16: np.ndarray) -> np.ndarray:
    """ensure_gray_uint8_from_bytes for a decoded uint16 image, server.py:509-523: H x W stays, H x W x {3,4} BGR(A) is
    gray-converted on 16 bits; then `>> 8`."""
    assert img16.dtype == np.uint16
    if img16.ndim == 3:
        img16 = bgr2gray_u16(img16)
    return u16_to_u8_shr8(img16)


def gray_u8_from_u16_bytes(raw: np.ndarray, w: int, channels: int, big_endian: bool = True, rgb_order: bool = True) -> np.ndarray:
    """The same for samples given as bytes (rows of 2 * w * channels bytes, e.g. reconstructed PNG scan lines: big-endian,
    R first): what ipx_gray_from_u16_u8 computes.  channels 2 = gray + alpha (cv2 decodes it to B = G = R = gray, A)."""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    h = raw.shape[-2]
    a = raw.reshape(raw.shape[:-1] + (w, channels, 2)).astype(np.uint16)
    s = (a[..., 0] << 8 | a[..., 1]) if big_endian else (a[..., 1] << 8 | a[..., 0])
    if channels < 3:
        return u16_to_u8_shr8(np.ascontiguousarray(s[..., 0]))
    bgr = s[..., [2, 1, 0]] if rgb_order else s[..., :3]
    lead = bgr.shape[:-3]
    out = u16_to_u8_shr8(bgr2gray_u16(np.ascontiguousarray(bgr.reshape((-1, w, 3)))))
    return out.reshape(lead + (h, w))


def u16_to_u8_shr8(img16: np.ndarray) -> np.ndarray:
    """16-bit -> 8-bit scaling `(img >> 8).astype(uint8)`, server.py:521-523."""
    assert img16.dtype == np.uint16
    return (img16 >> 8).astype(np.uint8)


# --------------------------------------------------------------------------
# LUT application (every pixel-wise transform of C4 is a 256-entry table)
# --------------------------------------------------------------------------
def lut_u8(img: np.ndarray, lut: np.ndarray) -> np.ndarray:
    assert img.dtype == np.uint8 and lut.dtype == np.uint8 and lut.shape == (256,)
    return lut[img]


# --------------------------------------------------------------------------
# A5  negative                                        server.py:375-376
# --------------------------------------------------------------------------
def negative_u8(img: np.ndarray) -> np.ndarray:
    return (255 - img).astype(np.uint8)


# --------------------------------------------------------------------------
# A7  cv2.threshold(img, T, 255, THRESH_BINARY)       server.py:378-381
# --------------------------------------------------------------------------
def threshold_u8(img: np.ndarray, T: int) -> np.ndarray:
    """OpenCV thresh.cpp, 8-bit path: ithresh = floor(T); ithresh < 0 -> all 255,
    ithresh >= 255 -> all 0, otherwise `v > ithresh ? 255 : 0` (strictly greater,
    README.MD:99)."""
    assert img.dtype == np.uint8
    return np.where(img.astype(np.int32) > int(np.floor(T)), 255, 0).astype(np.uint8)


# --------------------------------------------------------------------------
# A8  gamma / power-law                               server.py:362-373
# --------------------------------------------------------------------------
def gamma_u8(img: np.ndarray, gamma: float) -> np.ndarray:
    """Whole-image evaluation with exactly the reference's expressions (float32,
    truncating cast)."""
    r = img.astype(np.float32) / 255.0
    s = np.power(r, gamma)
    return np.clip((s * 255.0), 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------
# A9  log transform                                   server.py:383-388
# --------------------------------------------------------------------------
def log_u8(img: np.ndarray, c: float = 2.0) -> np.ndarray:
    r = img.astype(np.float32) / 255.0
    s = np.log1p(c * r)
    s /= (s.max() + 1e-12)
    return (s * 255.0).astype(np.uint8)


# --------------------------------------------------------------------------
# A10 piecewise-linear contrast stretch                server.py:443-469
# --------------------------------------------------------------------------
def piecewise_clamp_params(r1: int, s1: int, r2: int, s2: int):
    """Parameter clamping, server.py:454-457."""
    r1 = max(1, min(254, r1))
    r2 = max(r1 + 1, min(255, r2))
    s1 = int(np.clip(s1, 0, 255))
    s2 = int(np.clip(s2, 0, 255))
    return r1, s1, r2, s2


def piecewise_u8(img: np.ndarray, r1: int, s1: int, r2: int, s2: int) -> np.ndarray:
    r1, s1, r2, s2 = piecewise_clamp_params(r1, s1, r2, s2)
    x = img.astype(np.float32)
    y = np.zeros_like(x, dtype=np.float32)
    idx1 = x <= r1
    y[idx1] = (s1 / float(r1)) * x[idx1]
    idx2 = (x > r1) & (x <= r2)
    y[idx2] = s1 + (s2 - s1) * (x[idx2] - r1) / float(r2 - r1)
    idx3 = x > r2
    with np.errstate(divide="ignore", invalid="ignore"):
        # r2 == 255 divides an EMPTY float32 selection by 0.0: numpy warns, nothing is
        # raised and nothing is written.  Python floats stay "weak" (NEP 50) so the
        # arithmetic stays float32, as in the reference.
        y[idx3] = s2 + (255.0 - s2) * (x[idx3] - r2) / float(255 - r2)
    return np.clip(np.round(y), 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------
# A6  cv2.equalizeHist                                server.py:243-248
# --------------------------------------------------------------------------
def hist_u8(img: np.ndarray) -> np.ndarray:
    assert img.dtype == np.uint8
    return np.bincount(img.reshape(-1), minlength=256).astype(np.uint32)


def equalize_lut_from_hist(hist: np.ndarray) -> np.ndarray:
    """OpenCV histogram.cpp equalizeHist(): first non-zero bin i0; if it holds
    every pixel the output is the constant i0; else scale = 255.f/(total-hist[i0])
    in float32, lut[i0]=0, lut[i] = saturate_cast<uchar>(float(sum)*scale) with
    round-half-to-even (cvRound).  Entries below i0 are never read; set to 0."""
    hist = np.asarray(hist).astype(np.int64)
    total = int(hist.sum())
    lut = np.zeros(256, np.uint8)
    if total == 0:
        return lut
    i0 = int(np.flatnonzero(hist)[0])
    if hist[i0] == total:
        lut[:] = i0
        return lut
    scale = np.float32(255.0) / np.float32(np.int32(total - hist[i0]))
    csum = np.cumsum(hist[i0 + 1:]).astype(np.int32)
    v = np.rint(csum.astype(np.float32) * scale)          # rint = half-to-even
    lut[i0 + 1:] = np.clip(v, 0, 255).astype(np.uint8)
    return lut


def equalize_hist_u8(img: np.ndarray) -> np.ndarray:
    return equalize_lut_from_hist(hist_u8(img))[img]


# --------------------------------------------------------------------------
# borders: OpenCV BORDER_DEFAULT == BORDER_REFLECT_101  (gfedcb|abcdefgh|gfedcba)
# --------------------------------------------------------------------------
def _pad101(a: np.ndarray, r: int) -> np.ndarray:
    """Pad the two leading (spatial) axes by r with REFLECT_101.  np.pad's
    'reflect' is the same triangular extension as cv::borderInterpolate, also
    for r larger than the axis; a length-1 axis replicates (borderInterpolate
    returns 0)."""
    pads = [(r, r), (r, r)] + [(0, 0)] * (a.ndim - 2)
    out = a
    for ax in (0, 1):
        p = [(0, 0)] * a.ndim
        p[ax] = pads[ax]
        out = np.pad(out, p, mode="reflect" if out.shape[ax] > 1 else "edge")
    return out


# --------------------------------------------------------------------------
# A11 Laplacian sharpening                             server.py:263-277
# --------------------------------------------------------------------------
def laplacian_sharpen_u8(img: np.ndarray) -> np.ndarray:
    """`clip(f64(c) - Laplacian(c, CV_64F), 0, 255).astype(u8)` per channel.
    cv2.Laplacian ksize=1 -> kernel [[0,1,0],[1,-4,1],[0,1,0]], REFLECT_101;
    exact in integers: clip(5c - (n+s+e+w), 0, 255)."""
    assert img.dtype == np.uint8 and img.ndim in (2, 3)
    p = _pad101(img, 1).astype(np.int32)
    c = p[1:-1, 1:-1]
    out = 5 * c - (p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:])
    return np.clip(out, 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------
# A13 3x3 box blur and Sobel magnitude -- NOT in the reference (build's own)
# --------------------------------------------------------------------------
def box3_u8(img: np.ndarray) -> np.ndarray:
    """cv2.blur(img,(3,3)) convention: round-half-even(sum9 / 9), REFLECT_101.
    An integer sum over 9 is never a tie, so this is (sum + 4) // 9."""
    p = _pad101(img, 1).astype(np.int32)
    H, W = img.shape[:2]
    s = sum(p[dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3))
    return ((s + 4) // 9).astype(np.uint8)


def sobel_u8(img: np.ndarray) -> np.ndarray:
    """gx = [[-1,0,1],[-2,0,2],[-1,0,1]], gy = gx^T, REFLECT_101,
    out = min(255, |gx| + |gy|)."""
    p = _pad101(img, 1).astype(np.int32)
    H, W = img.shape[:2]
    w = lambda dy, dx: p[dy:dy + H, dx:dx + W]
    gx = (w(0, 2) + 2 * w(1, 2) + w(2, 2)) - (w(0, 0) + 2 * w(1, 0) + w(2, 0))
    gy = (w(2, 0) + 2 * w(2, 1) + w(2, 2)) - (w(0, 0) + 2 * w(0, 1) + w(0, 2))
    return np.minimum(255, np.abs(gx) + np.abs(gy)).astype(np.uint8)


# --------------------------------------------------------------------------
# A12 cv2.GaussianBlur(img, (k,k), 0) on uint8          server.py:427-428
# --------------------------------------------------------------------------
_SMALL_GAUSS = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
                7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125],
                9: [4 / 256, 13 / 256, 30 / 256, 51 / 256, 60 / 256, 51 / 256, 30 / 256, 13 / 256, 4 / 256]}


def gauss_kernel_float(k: int, sigma: float = 0.0) -> np.ndarray:
    """OpenCV smooth.dispatch.cpp getGaussianKernelBitExact(), operation by operation: fixed tables for sigma<=0 and
    k in {1,3,5,7,9}; else sigma = mulAdd(k, 0.15, 0.35) (ONE rounding), scale2X = -0.125 / sigma^2 with the integer
    abscissa x = 2 i - (k - 1), taps t_i = exp(x*x * scale2X) for the FIRST half only, sum = 2 * sum(t_i) + 1 (the centre
    tap is exp(0) = 1), every tap multiplied by mul1 = 1 / sum.  OpenCV evaluates exp in softfloat, float64 libm here;
    tests/test_oracle_golden.py proves with 60-digit arithmetic that no Q0.8 tap of k <= 255 is near enough to a
    rounding tie for an ulp of difference to show."""
    assert k >= 1 and k % 2 == 1
    if sigma <= 0 and k in _SMALL_GAUSS:
        return np.array(_SMALL_GAUSS[k], np.float64)
    from fractions import Fraction
    # a fused multiply-add is the exact rational result rounded once
    sg = float(sigma) if sigma > 0 else float(Fraction(k) * Fraction(0.15) + Fraction(0.35))
    scale2 = -0.125 / (sg * sg)
    n2 = (k - 1) // 2
    vals = [math.exp(float((2 * i - (k - 1)) ** 2) * scale2) for i in range(n2)]
    tot = 0.0
    for t in vals:
        tot += t
    tot = tot * 2.0 + 1.0
    mul1 = 1.0 / tot
    out = np.empty(k, np.float64)
    for i in range(n2):
        out[i] = out[k - 1 - i] = vals[i] * mul1
    out[n2] = 1.0 * mul1
    return out


def gauss_kernel_q8(k: int, sigma: float = 0.0) -> np.ndarray:
    """getGaussianKernelFixedPoint_ED(): Q0.8 taps with error diffusion from the
    ends inward; centre tap = 256 - 2*sum(others) so the sum is exactly 256."""
    kf = gauss_kernel_float(k, sigma)
    out = np.zeros(k, np.int64)
    err = 0.0
    for i in range(k // 2):
        adj = kf[i] * 256.0 + err
        v0 = int(np.rint(adj))                      # cvRound
        err = adj - v0
        out[i] = out[k - 1 - i] = v0
    out[k // 2] = 256 - 2 * int(out[:k // 2].sum())
    return out.astype(np.uint16)


def sep_filter_q8_u8(img: np.ndarray, kx: np.ndarray, ky: np.ndarray) -> np.ndarray:
    """OpenCV fixed-point separable smoothing (smooth.simd.hpp, ufixedpoint16/32):
    row pass u8 x Q0.8 -> Q8.8 (exact), column pass -> Q8.16, out = (acc + 2^15) >> 16.
    Interleaved channels are filtered independently.  REFLECT_101."""
    assert img.dtype == np.uint8
    kx = np.asarray(kx, np.int64)
    ky = np.asarray(ky, np.int64)
    rx, ry = len(kx) // 2, len(ky) // 2
    H, W = img.shape[:2]
    a = img.astype(np.int64)
    # horizontal
    p = a
    pads = [(0, 0)] * a.ndim
    pads[1] = (rx, rx)
    p = np.pad(a, pads, mode="reflect" if W > 1 else "edge") if rx else a
    h = sum(int(kx[j]) * p[:, j:j + W] for j in range(len(kx)))
    pads = [(0, 0)] * a.ndim
    pads[0] = (ry, ry)
    q = np.pad(h, pads, mode="reflect" if H > 1 else "edge") if ry else h
    v = sum(int(ky[i]) * q[i:i + H] for i in range(len(ky)))
    return np.clip((v + (1 << 15)) >> 16, 0, 255).astype(np.uint8)


def gaussian_blur_u8(img: np.ndarray, k: int, sigma: float = 0.0) -> np.ndarray:
    q = gauss_kernel_q8(k, sigma)
    return sep_filter_q8_u8(img, q, q)


def gaussian_blur_float_ref(img: np.ndarray, k: int) -> np.ndarray:
    """float64 Gaussian with the same sigma rule, for the '+-1 LSB vs a floating
    filter' report (hard part 2).  Not an oracle."""
    kf = gauss_kernel_float(k, 0.0)
    r = k // 2
    H, W = img.shape[:2]
    p = _pad101(img, r).astype(np.float64)
    h = sum(kf[j] * p[:, j:j + W] for j in range(k))
    v = sum(kf[i] * h[i:i + H] for i in range(k))
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------
# (f)-1  Lee filter                                    server.py:711-732
# --------------------------------------------------------------------------
def lee_filter_u8(img: np.ndarray, window_size: int = 5) -> np.ndarray:
    """Vectorised restatement of the per-pixel loop: float32 window mean/var
    (numpy pairwise sums over k*k float32 values), interior only, borders
    copied."""
    H, W = img.shape
    out = img.copy().astype(np.float32)
    k = window_size // 2
    if H - 2 * k <= 0 or W - 2 * k <= 0:
        return np.clip(out, 0, 255).astype(np.uint8)
    from numpy.lib.stride_tricks import sliding_window_view
    win = sliding_window_view(img, (2 * k + 1, 2 * k + 1)).astype(np.float32)
    win = np.ascontiguousarray(win).reshape(H - 2 * k, W - 2 * k, -1)
    ctr = img[k:H - k, k:W - k]
    res = np.empty((H - 2 * k, W - 2 * k), np.float32)
    for y in range(win.shape[0]):                      # row loop keeps memory small
        w = win[y]
        mean_w = w.mean(axis=1)
        var_w = w.var(axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            k_lee = var_w / (var_w + mean_w ** 2 / (mean_w + 1e-12))
            r = mean_w + k_lee * (ctr[y] - mean_w)
        res[y] = np.where(var_w > 0, r, mean_w)
    out[k:H - k, k:W - k] = res
    return np.clip(out, 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------
# (f)-2  denoise_manual histogram statistics            server.py:190-211
# --------------------------------------------------------------------------
def denoise_stats(img: np.ndarray, method_name: str = "median") -> dict:
    H, W = img.shape
    total = float(H * W)
    n_black = int(np.sum(img == 0))
    n_white = int(np.sum(img == 255))
    extreme_ratio = float((n_black + n_white) / total)
    flat = img.flatten().astype(np.float32)
    mean_val = float(np.mean(flat))
    var = float(np.var(flat))
    cv_val = float(np.std(flat) / (mean_val + 1e-12))
    if extreme_ratio > 0.02:
        s = "median"
    elif cv_val > 0.28:
        s = "lee"
    elif 100 < var < 2000:
        s = "bilateral"
    else:
        s = method_name
    return {"extreme_ratio": extreme_ratio, "mean": mean_val, "variance": var, "cv": cv_val,
            "suggested_method": s}


def pairwise_sum_f32(a: np.ndarray) -> np.float32:
    """What np.add.reduce does to a contiguous float32 vector, spelled out (numpy 2.2, default bufsize): buffers of
    8192 elements are summed PAIRWISE (blocks of <= 128 values with 8 interleaved accumulators, larger inputs halved at
    n/2 rounded down to a multiple of 8) and added to the running total in sequence.  The GPU statistics kernel
    (ipx_stats.hip) follows this order; tests compare this function with np.add.reduce itself."""
    a = np.ascontiguousarray(a, np.float32)

    def block(v):
        n = len(v)
        if n < 8:
            res = np.float32(0)
            for x in v:
                res = np.float32(res + x)
            return res
        r = v[:8].copy()
        m = n - n % 8
        for i in range(8, m, 8):
            r = r + v[i:i + 8]
        res = np.float32(np.float32(np.float32(r[0] + r[1]) + np.float32(r[2] + r[3])) +
                         np.float32(np.float32(r[4] + r[5]) + np.float32(r[6] + r[7])))
        for x in v[m:]:
            res = np.float32(res + x)
        return res

    def pw(v):
        n = len(v)
        if n <= 128:
            return block(v)
        n2 = n // 2
        n2 -= n2 % 8
        return np.float32(pw(v[:n2]) + pw(v[n2:]))

    res = np.float32(0)
    for o in range(0, len(a), 8192):
        res = np.float32(res + pw(a[o:o + 8192]))
    return res


# --------------------------------------------------------------------------
# (f)-4  median / morphology of the `median` pipeline     server.py:215, 426-434
#        [OCV-MEM] definitions; parity unpinned (OpenCV absent)
# --------------------------------------------------------------------------
def median_blur_u8(img: np.ndarray, k: int) -> np.ndarray:
    """cv2.medianBlur: BORDER_REPLICATE, the (k*k-1)/2-th smallest value of the k x k window."""
    r = k // 2
    p = np.pad(img, r, mode="edge")
    from numpy.lib.stride_tricks import sliding_window_view
    win = sliding_window_view(p, (k, k)).reshape(img.shape[0], img.shape[1], k * k)
    return np.partition(win, (k * k - 1) // 2, axis=2)[:, :, (k * k - 1) // 2].astype(np.uint8)


def ellipse_se(m: int) -> np.ndarray:
    """cv2.getStructuringElement(MORPH_ELLIPSE, (m, m)): row i spans [c-dx, c+dx] with
    dx = cvRound(c*sqrt((r*r-dy*dy)/(r*r)))."""
    r = c = m // 2
    se = np.zeros((m, m), np.uint8)
    for i in range(m):
        dy = i - r
        dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * (1.0 / (r * r) if r else 0.0))))
        se[i, max(c - dx, 0):min(c + dx + 1, m)] = 1
    return se


def _morph(img: np.ndarray, se: np.ndarray, dilate: bool) -> np.ndarray:
    kh, kw = se.shape
    ry, rx = kh // 2, kw // 2
    fill = 0 if dilate else 255                     # out-of-image pixels never win
    p = np.pad(img, ((ry, ry), (rx, rx)), mode="constant", constant_values=fill)
    out = np.full(img.shape, fill, np.uint8)
    for i in range(kh):
        for j in range(kw):
            if se[i, j]:
                v = p[i:i + img.shape[0], j:j + img.shape[1]]
                out = np.maximum(out, v) if dilate else np.minimum(out, v)
    return out


def erode_u8(img, se): return _morph(img, np.asarray(se, np.uint8), False)
def dilate_u8(img, se): return _morph(img, np.asarray(se, np.uint8), True)
def morph_open_u8(img, se): return dilate_u8(erode_u8(img, se), se)
def morph_close_u8(img, se): return erode_u8(dilate_u8(img, se), se)


def median_pipeline_u8(img: np.ndarray, k: int, g: int, morph: int, passes: int) -> np.ndarray:
    """server.py:426-434 (without the optional NLM step)."""
    if g > 0:
        img = gaussian_blur_u8(img, g)
    for _ in range(passes):
        img = median_blur_u8(img, k)
    if morph > 0:
        se = ellipse_se(morph)
        img = morph_close_u8(morph_open_u8(img, se), se)
    return img


# --------------------------------------------------------------------------
# pipelines BASELINE.json names (used by tests and bench cpu_baseline)
# --------------------------------------------------------------------------
def pipeline_cfg1(bgr: np.ndarray) -> np.ndarray:
    """configs[0]: grayscale -> 3x3 box blur -> histogram equalization."""
    return equalize_hist_u8(box3_u8(bgr2gray_u8(bgr)))


def pipeline_cfg2(bgr: np.ndarray):
    """configs[1]: 5x5 Gaussian (per channel), Sobel on gray, hist-eq on gray."""
    g = bgr2gray_u8(bgr)
    return gaussian_blur_u8(bgr, 5), sobel_u8(g), equalize_hist_u8(g)

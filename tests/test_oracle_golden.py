"""CPU suite: the oracle against the golden vectors made from the real server.py, the two restatements
against each other, the host LUT builders against both, and known-answer anchors."""
import numpy as np
import pytest

import oracle as O
import coracle as CO
import ipx.luts as luts


def _expected(O, img, mode, form):
    if mode in ("gamma", "powerlaw", "power-law"):
        return O.gamma_u8(img, float(form.get("gamma") or form.get("gamma_value") or form.get("g")))
    if mode == "negative":
        return O.negative_u8(img)
    if mode in ("log", "logarithmic", "logrithmic"):
        return O.log_u8(img, float(form.get("c", "2.0")))
    if mode in ("piecewise", "piecewise_linear", "stretch"):
        return O.piecewise_u8(img, *[int(form[k]) for k in ("r1", "s1", "r2", "s2")])
    if mode == "denoise_manual":
        return O.lee_filter_u8(img, _lee_kernel(form))
    return img          # gray8 / default / unknown mode on a gray input


def _lee_kernel(form):
    k = int(form.get("kernel", "5"))                       # server.py:185-188
    k += 1 if k % 2 == 0 else 0
    return max(3, min(31, k))


def _lut_for(mode, form, img):
    if mode in ("gamma", "powerlaw", "power-law"):
        return luts.gamma(float(form.get("gamma") or form.get("gamma_value") or form.get("g")))
    if mode == "negative":
        return luts.negative()
    if mode in ("log", "logarithmic", "logrithmic"):
        return luts.log(float(form.get("c", "2.0")), np.bincount(img.ravel(), minlength=256))
    if mode in ("piecewise", "piecewise_linear", "stretch"):
        return luts.piecewise(*luts.piecewise_clamp(*[int(form[k]) for k in ("r1", "s1", "r2", "s2")]))
    return np.arange(256, dtype=np.uint8)


def test_oracle_matches_real_server_outputs(golden):
    g, meta = golden
    n = 0
    for c in meta["cases"]:
        if c["status"] != 200:
            continue
        img, out = g["in_" + c["input"]], g["out_" + c["name"]]
        np.testing.assert_array_equal(_expected(O, img, c["mode"] or "gray8", c["form"]), out, err_msg=c["name"])
        n += 1
    assert n >= 60


def test_host_luts_match_real_server_outputs(golden):
    """LUT built on 256 levels, then applied == the reference's whole-image evaluation (bit-exact)."""
    g, meta = golden
    for c in meta["cases"]:
        if c["status"] != 200 or c["mode"] == "denoise_manual":
            continue
        img, out = g["in_" + c["input"]], g["out_" + c["name"]]
        lut = _lut_for(c["mode"] or "gray8", c["form"], img)
        np.testing.assert_array_equal(lut[img], out, err_msg=c["name"])


def test_lee_filter_matches_real_server(golden):
    g, _ = golden
    for w in (3, 5, 7):
        np.testing.assert_array_equal(O.lee_filter_u8(g["in_lee"], w), g[f"out_lee_{w}"])
    for w in (9, 15, 31):
        np.testing.assert_array_equal(O.lee_filter_u8(g["in_rand"], w), g[f"out_lee_rand_{w}"])
    # windows where a fused multiply-add anywhere in the chain would change the result by one
    np.testing.assert_array_equal(O.lee_filter_u8(g["in_lee_rounding"], 3), g["out_lee_rounding_3"])


def test_denoise_stats_match_real_server(golden):
    """server.py:190-211 as the real server reported them (rounded in its JSON): the oracle (numpy float32, as the
    reference) reproduces them exactly; the product's histogram form (float64 on exact counts) to float32 rounding."""
    import ipx.ops as real_ops                              # stats_from_hist is host arithmetic on 256 counts
    g, meta = golden
    n = 0
    for c in meta["cases"]:
        if c["mode"] != "denoise_manual" or c["status"] != 200:
            continue
        img, ref = g["in_" + c["input"]], c["params"]["histogram"]
        s = O.denoise_stats(img, "lee")
        assert (round(s["extreme_ratio"], 4), round(s["variance"], 2), round(s["cv"], 4), s["suggested_method"]) == \
               (ref["extreme_ratio"], ref["variance"], ref["cv"], ref["suggested_method"]), c["name"]
        p = real_ops.stats_from_hist(np.bincount(img.ravel(), minlength=256), "lee")
        assert p["suggested_method"] == ref["suggested_method"]
        assert abs(p["variance"] - s["variance"]) <= 2e-6 * max(1.0, s["variance"])
        assert abs(p["cv"] - s["cv"]) <= 2e-6 and abs(p["extreme_ratio"] - s["extreme_ratio"]) <= 1e-12
        n += 1
    assert n >= 7


def test_piecewise_probe_from_survey():
    lut = luts.piecewise(*luts.piecewise_clamp(70, 10, 180, 245))
    assert [int(lut[i]) for i in (0, 70, 71, 180, 181, 255)] == [0, 10, 12, 245, 245, 255]


def test_threshold_semantics():
    img = np.arange(256, dtype=np.uint8).reshape(16, 16)
    assert np.array_equal(O.threshold_u8(img, 128), np.where(img > 128, 255, 0))
    assert O.threshold_u8(img, -1).min() == 255 and O.threshold_u8(img, 255).max() == 0
    for T in (-5, 0, 1, 127, 128, 254, 255, 300):
        np.testing.assert_array_equal(luts.threshold(T)[img], O.threshold_u8(img, T))


@pytest.mark.parametrize("shape", [(37, 53, 3), (20, 31), (1, 7, 3), (5, 1), (2, 2, 3), (33, 70, 4), (3, 3)])
def test_c_and_numpy_restatements_agree(shape):
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    if img.ndim == 3:
        np.testing.assert_array_equal(O.bgr2gray_u8(img), CO.bgr2gray_u8(img))
    np.testing.assert_array_equal(O.laplacian_sharpen_u8(img), CO.laplacian_sharpen_u8(img))
    np.testing.assert_array_equal(O.box3_u8(img), CO.box3_u8(img))
    np.testing.assert_array_equal(O.sobel_u8(img), CO.sobel_u8(img))
    for k in (1, 3, 5, 7, 9, 15, 31):
        np.testing.assert_array_equal(O.gauss_kernel_q8(k), CO.gauss_kernel_q8(k))
        np.testing.assert_array_equal(O.gaussian_blur_u8(img, k), CO.gaussian_blur_u8(img, k))
    gray = img if img.ndim == 2 else np.ascontiguousarray(img[..., 0])
    np.testing.assert_array_equal(O.hist_u8(gray), CO.hist_u8(gray))
    np.testing.assert_array_equal(O.equalize_hist_u8(gray), CO.equalize_hist_u8(gray))


def test_gauss_kernels_known_answers():
    assert O.gauss_kernel_q8(3).tolist() == [64, 128, 64]
    assert O.gauss_kernel_q8(5).tolist() == [16, 64, 96, 64, 16]
    assert O.gauss_kernel_q8(7).tolist() == [8, 28, 56, 72, 56, 28, 8]
    for k in range(1, 64, 2):
        q = O.gauss_kernel_q8(k)
        assert int(q.sum()) == 256 and np.array_equal(q, q[::-1])


def test_identity_anchors():
    """constant images are fixed points of sharpen / equalize / Gaussian / box."""
    for v in (0, 77, 255):
        c = np.full((9, 11), v, np.uint8)
        c3 = np.full((9, 11, 3), v, np.uint8)
        for f in (O.laplacian_sharpen_u8, O.box3_u8, O.equalize_hist_u8, lambda a: O.gaussian_blur_u8(a, 5),
                  lambda a: O.gaussian_blur_u8(a, 31)):
            np.testing.assert_array_equal(f(c), c)
        np.testing.assert_array_equal(O.laplacian_sharpen_u8(c3), c3)
        assert O.sobel_u8(c).max() == 0
        np.testing.assert_array_equal(O.bgr2gray_u8(c3), c)       # coefficients sum to 2^15


def test_equalize_uniform_levels_known_answer():
    N = 256 * 40
    img = np.repeat(np.arange(256, dtype=np.uint8), 40).reshape(80, 128)
    lut = O.equalize_lut_from_hist(O.hist_u8(img))
    exp = np.rint(np.arange(256) * (N / 256) * np.float32(255.0) / np.float32(N - N / 256))
    exp[0] = 0
    np.testing.assert_array_equal(lut, exp.astype(np.uint8))


def test_equalize_lut_float32_at_2_30_pixels():
    rng = np.random.default_rng(3)
    p = rng.random(256)
    hist = np.floor(p / p.sum() * (1 << 30)).astype(np.int64)
    hist[255] += (1 << 30) - hist.sum()
    np.testing.assert_array_equal(O.equalize_lut_from_hist(hist), CO.equalize_lut_from_hist(hist.astype(np.uint32)))


def test_gaussian_fixed_point_vs_float_within_1lsb_small_k():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (40, 44), dtype=np.uint8)
    for k in (3, 5, 7):
        d = np.abs(O.gaussian_blur_u8(img, k).astype(int) - O.gaussian_blur_float_ref(img, k).astype(int))
        assert d.max() <= 1


def test_denoise_stats_thresholds():
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    s = O.denoise_stats(img)
    assert s["suggested_method"] in ("median", "lee", "bilateral")
    sp = np.full((64, 64), 128, np.uint8)
    sp[::4, ::4] = 0
    assert O.denoise_stats(sp)["suggested_method"] == "median"


def test_median_and_morphology_anchors():
    """Known answers that need no OpenCV: OpenCV's documented 3x3 / 5x5 elliptical elements, median of constant and of
    a salt-and-pepper speck, opening removes an isolated bright pixel, closing an isolated dark one."""
    assert O.ellipse_se(1).tolist() == [[1]]
    assert O.ellipse_se(3).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    assert O.ellipse_se(5).tolist() == [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]
    img = np.full((9, 9), 100, np.uint8)
    assert np.array_equal(O.median_blur_u8(img, 5), img)
    img[4, 4] = 255
    assert np.array_equal(O.median_blur_u8(img, 3), np.full((9, 9), 100, np.uint8))
    se = O.ellipse_se(3)
    assert np.array_equal(O.morph_open_u8(img, se), np.full((9, 9), 100, np.uint8))
    img[4, 4] = 0
    assert np.array_equal(O.morph_close_u8(img, se), np.full((9, 9), 100, np.uint8))
    ramp = np.arange(81, dtype=np.uint8).reshape(9, 9)
    assert O.median_blur_u8(ramp, 3)[0, 0] == np.sort([0, 0, 1, 0, 0, 1, 9, 9, 10])[4]      # BORDER_REPLICATE corner
    assert O.erode_u8(ramp, np.ones((3, 3), np.uint8))[0, 0] == 0 and O.dilate_u8(ramp, np.ones((3, 3), np.uint8))[8, 8] == 80


def test_numpy_reduction_order_is_what_the_stats_kernel_assumes():
    """ipx_stats.hip reproduces np.add.reduce's float32 order (8192-element buffers, pairwise inside): pin that order
    against this numpy, on data whose float32 sum depends on it."""
    rng = np.random.default_rng(21)
    for n in (1, 7, 9, 128, 129, 961, 8191, 8192, 8193, 20000, 123457, 300 * 411):
        a = (rng.random(n) * 255).astype(np.float32)
        assert O.pairwise_sum_f32(a) == np.add.reduce(a), n
        b = rng.integers(0, 256, n).astype(np.float32)
        assert O.pairwise_sum_f32(b * b) == np.add.reduce(b * b), n
    big = rng.integers(0, 256, 1080 * 1920, dtype=np.uint8).astype(np.float32)     # sums beyond 2^24: rounding order matters
    assert O.pairwise_sum_f32(big) == np.add.reduce(big) == np.float32(np.mean(big) * 0 + np.add.reduce(big))


def test_gauss_taps_are_far_from_rounding_ties_for_every_k():
    """Pins the Q0.8 tap tables WITHOUT OpenCV: for k = 11..255 the taps are recomputed in 60-digit
    decimal arithmetic (exact sigma = 0.15 k + 0.35, exp, normalisation, error diffusion) and every value handed to
    cvRound, f[i] * 256 + err, must be farther from a .5 tie than any float64 / softfloat evaluation error could reach
    (margin 1e-9 against an error below 1e-12).  Then the order of float operations -- fused or unfused sigma, full or
    half sum, divide or multiply by the reciprocal, libm or softfloat exp -- cannot change a tap, and the float64
    restatements (numpy, C, and the library's host helper) must reproduce the high-precision table."""
    import ctypes as C
    from decimal import Decimal, getcontext
    import ipx._lib as L
    getcontext().prec = 60
    lib = L.load()
    worst = Decimal(1)
    for k in range(11, 256, 2):
        sg = Decimal(0.15) * k + Decimal(0.35)                  # the double constants OpenCV holds, combined exactly
        scale2 = Decimal(-0.125) / (sg * sg)
        n2 = (k - 1) // 2
        vals = [(Decimal((2 * i - (k - 1)) ** 2) * scale2).exp() for i in range(n2)]
        tot = sum(vals) * 2 + 1
        err = Decimal(0)
        taps = [0] * k
        for i in range(n2):
            adj = vals[i] / tot * 256 + err
            v0 = int(adj.to_integral_value(rounding="ROUND_HALF_EVEN"))
            margin = abs(abs(adj - v0) - Decimal("0.5"))
            worst = min(worst, margin)
            err = adj - v0
            taps[i] = taps[k - 1 - i] = v0
        taps[n2] = 256 - 2 * sum(taps[:n2])
        assert worst > Decimal("1e-9"), (k, worst)
        assert O.gauss_kernel_q8(k).tolist() == taps, k
        assert CO.gauss_kernel_q8(k).tolist() == taps, k
        out = np.zeros(k, np.uint16)
        assert lib.ipx_gauss_kernel_q8(k, C.c_double(0.0), out.ctypes.data) == 0
        assert out.tolist() == taps, k
        assert min(taps) >= 0 and sum(taps) == 256


def test_markstein_division_by_window_size(tmp_path):
    """ipx_lee.hip divides the variance sum by n = 9 / 25 / 49 / 81 / 121 with y = fl(a c), r = fma(-n, y, a), q = fma(r, c, y), c = fl(1 / n).
    The three steps scale exactly with powers of two, so correct rounding for all 2^23 significands of three binades (and a
    strided sample of fifty more) is correct rounding for every float whose quotient is normal.  Plain C, gcc, no GPU."""
    import shutil, subprocess
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    src = tmp_path / "mk.c"
    src.write_text(r"""
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
int main(void) {
    const int ns[] = {9, 25, 49, 81, 121};
    long bad = 0;
    for (int ni = 0; ni < 5; ++ni) {
        const float n = (float)ns[ni], c = 1.0f / n;
        for (int e = 90; e <= 150; ++e) {
            const uint32_t step = (e >= 126 && e <= 128) ? 1u : 4099u;
            for (uint32_t m = 0; m < (1u << 23); m += step) {
                uint32_t bits = ((uint32_t)e << 23) | m; float a; memcpy(&a, &bits, 4);
                const float y = a * c, r = fmaf(-n, y, a), q = fmaf(r, c, y);
                if (q != a / n) ++bad;
            }
        }
    }
    printf("%ld\n", bad);
    return 0;
}
""")
    exe = tmp_path / "mk"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"], check=True)
    assert subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip() == "0"


def test_lee_division_sequence_on_the_operands_the_kernel_feeds_it(tmp_path):
    """ipx_lee.hip's div_safe_range(a, b) = the hardware division's refinement steps without its range handling.  It is used for
    the gain var / (var + q) (a <= b) AND, in the 7 x 7 ... 11 x 11 forms without a table, for q = fl(mean^2) / fl(mean + 1e-12),
    where a reaches 65025 against b <= 255 (quotient up to 255).  For every window sum S > 0 of n = 49, 81, 121 values the
    sequence must give fl(a / b) -- also when the reciprocal seed (v_rcp_f32: 1 ulp) is off by one unit either way -- and for a
    sample of gains.  Plain C with fmaf, gcc, no GPU."""
    import shutil, subprocess
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    src = tmp_path / "dv.c"
    src.write_text(r"""
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
static float seq(float a, float b, int ulp) {
    float r = 1.0f / b;
    uint32_t u; memcpy(&u, &r, 4); u += ulp; memcpy(&r, &u, 4);
    r = fmaf(fmaf(-b, r, 1.0f), r, r);
    float q = a * r;
    q = fmaf(fmaf(-b, q, a), r, q);
    return fmaf(fmaf(-b, q, a), r, q);
}
int main(void) {
    const int ns[] = {49, 81, 121};
    long bad = 0, n_checked = 0;
    for (int ni = 0; ni < 3; ++ni) {
        const float n = (float)ns[ni];
        for (int S = 1; S <= 255 * ns[ni]; ++S) {
            const float mean = (float)S / n, a = mean * mean, b = mean + 1e-12f;
            for (int ulp = -1; ulp <= 1; ++ulp) { if (seq(a, b, ulp) != a / b) ++bad; ++n_checked; }
            /* gains with this q and a sweep of variances (var = t / n for integers t, as the kernel forms them) */
            const float q = a / b;
            for (int t = 1; t < 4000000; t = t * 3 + 1) {
                const float var = (float)t / n;
                for (int ulp = -1; ulp <= 1; ++ulp) { if (seq(var, var + q, ulp) != var / (var + q)) ++bad; ++n_checked; }
            }
        }
    }
    printf("%ld %ld\n", bad, n_checked);
    return 0;
}
""")
    exe = tmp_path / "dv"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"], check=True)
    bad, checked = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert bad == "0" and int(checked) > 2_000_000

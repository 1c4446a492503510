"""The oracle's restatements of the OpenCV-backed operators against an INDEPENDENT implementation of their published
definitions (scipy.ndimage; OpenCV itself is not importable here and the reference holds no fixtures for these operators,
so their parity with cv2 stays "unpinned" -- DESIGN.md section 3).  What this pins: window shapes, border rules
(REFLECT_101 = scipy 'mirror', REPLICATE = 'nearest', morphology's ignore-outside rule), kernel orientation and
saturation -- the places a restatement can silently go wrong.  What it cannot pin: OpenCV's own rounding choices
(fixed-point Gaussian taps, cvRound), which the oracle takes from OpenCV's sources as cited in oracle/oracle.py."""
import numpy as np
import pytest
from scipy import ndimage

import oracle as O


def images():
    rng = np.random.default_rng(20240611)
    yield rng.integers(0, 256, (37, 53), dtype=np.uint8)
    yield rng.integers(0, 256, (8, 9), dtype=np.uint8)
    yield (rng.integers(0, 2, (40, 41)) * 255).astype(np.uint8)                        # two levels: every window is full of ties
    yy, xx = np.mgrid[0:33, 0:47]
    yield ((xx * 7 + yy * 13) % 256).astype(np.uint8)                                  # ramps: orientation shows


@pytest.mark.parametrize("k", [3, 5, 7, 9, 15])
def test_median_against_scipy(k):
    for img in images():
        np.testing.assert_array_equal(O.median_blur_u8(img, k), ndimage.median_filter(img, size=k, mode="nearest"))


@pytest.mark.parametrize("m", [3, 5, 7, 9])
def test_morphology_against_scipy(m):
    se = O.ellipse_se(m)
    assert se[m // 2].all() and se[:, m // 2].all() and (se == se[::-1]).all() and (se == se[:, ::-1]).all()
    asym = np.zeros((3, 5), np.uint8)
    asym[0, :2] = 1
    asym[1, 2] = 1
    asym[2, 4] = 1
    for img in images():
        for fp in (se, np.ones((m, m), np.uint8), asym):
            # cv2.erode / dilate anchor the element at its centre and do NOT reflect it (correlation, like scipy's footprint
            # for erosion; scipy's grey_dilation reflects the footprint, so it gets the mirrored one)
            np.testing.assert_array_equal(O.erode_u8(img, fp), ndimage.grey_erosion(img, footprint=fp, mode="constant", cval=255))
            np.testing.assert_array_equal(O.dilate_u8(img, fp), ndimage.grey_dilation(img, footprint=fp[::-1, ::-1], mode="constant", cval=0))
        np.testing.assert_array_equal(O.morph_open_u8(img, se),
                                      ndimage.grey_dilation(ndimage.grey_erosion(img, footprint=se, mode="constant", cval=255),
                                                            footprint=se, mode="constant", cval=0))


def test_sobel_laplacian_box_against_scipy():
    gx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
    lap = np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]])
    for img in images():
        f = img.astype(np.int64)
        sx = ndimage.correlate(f, gx, mode="mirror")
        sy = ndimage.correlate(f, gx.T, mode="mirror")
        np.testing.assert_array_equal(O.sobel_u8(img), np.minimum(255, np.abs(sx) + np.abs(sy)).astype(np.uint8))
        np.testing.assert_array_equal(O.laplacian_sharpen_u8(img), np.clip(f - ndimage.correlate(f, lap, mode="mirror"), 0, 255).astype(np.uint8))
        box = ndimage.correlate(f, np.ones((3, 3), np.int64), mode="mirror")
        np.testing.assert_array_equal(O.box3_u8(img), np.rint(box / 9.0).astype(np.uint8))
    bgr = np.random.default_rng(5).integers(0, 256, (20, 31, 3), dtype=np.uint8)       # per channel, channels never mix
    for c in range(3):
        np.testing.assert_array_equal(O.laplacian_sharpen_u8(bgr)[:, :, c], O.laplacian_sharpen_u8(np.ascontiguousarray(bgr[:, :, c])))


@pytest.mark.parametrize("k", [3, 5, 7, 9, 15, 31])
def test_gaussian_close_to_a_float_filter(k):
    """The Q0.8 fixed-point filter against a float64 separable filter (scipy correlate1d, REFLECT_101 = 'mirror') with the
    taps OpenCV documents for ksize <= 7 and sigma = 0, and exp(-x^2 / 2 sigma^2) normalised with
    sigma = 0.3 ((k - 1) / 2 - 1) + 0.8 beyond.  Tolerance stated: 1 grey level where the Q0.8 taps are the float taps
    exactly (k <= 7: two roundings against one), 2 levels beyond (the Q0.8 taps sum to exactly 256 and each is within
    1.5 / 512 of its float value; a black-and-white image shows the difference)."""
    small = {3: [0.25, 0.5, 0.25], 5: [1 / 16, 4 / 16, 6 / 16, 4 / 16, 1 / 16],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if k in small:
        taps = np.array(small[k])
    else:
        sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
        x = np.arange(k) - (k - 1) / 2
        taps = np.exp(-x * x / (2 * sigma * sigma))
        taps /= taps.sum()
    q = O.gauss_kernel_q8(k)
    assert int(q.sum()) == 256 and np.abs(q / 256.0 - taps).max() <= 1.5 / 512 and (q == q[::-1]).all()
    for img in images():
        if min(img.shape) <= k // 2:
            continue
        f = img.astype(np.float64)
        ref = ndimage.correlate1d(ndimage.correlate1d(f, taps, axis=1, mode="mirror"), taps, axis=0, mode="mirror")
        d = np.abs(O.gaussian_blur_u8(img, k).astype(np.int64) - np.rint(ref).astype(np.int64))
        assert d.max() <= (1 if k <= 7 else 2), (k, img.shape, d.max())


def test_gray_and_equalize_against_their_definitions():
    rng = np.random.default_rng(11)
    bgr = rng.integers(0, 256, (64, 80, 3), dtype=np.uint8)
    y = 0.114 * bgr[:, :, 0].astype(np.float64) + 0.587 * bgr[:, :, 1] + 0.299 * bgr[:, :, 2]
    assert np.abs(O.bgr2gray_u8(bgr).astype(np.float64) - y).max() <= 0.5 + 0.03       # Q14 weights (3 x 255 / 2^15 off at most), rounded to nearest
    g = np.clip(rng.normal(120, 30, (90, 70)), 0, 255).astype(np.uint8)
    eq = O.equalize_hist_u8(g)
    h = np.bincount(g.ravel(), minlength=256)
    cdf = np.cumsum(h)
    i0 = int(np.nonzero(h)[0][0])
    want = np.rint((cdf - cdf[i0]) * 255.0 / (g.size - h[i0])).clip(0, 255)
    assert np.abs(eq.astype(np.int64) - want[g].astype(np.int64)).max() <= 1             # float32 scale in OpenCV vs float64 here
    assert eq.min() == 0 and eq.max() == 255

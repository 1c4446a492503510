"""GPU parity: every libipx entry point (through the ctypes C-ABI) against the CPU oracle, bit-exact.

Bar: all of this path is integer / table arithmetic (the Gaussian is OpenCV's fixed-point one), so every
comparison is exact equality; the only float32 step (equalizeHist's scale) is compared exactly too.
"""
import numpy as np
import pytest
import torch

import oracle as O
import ipx.luts as luts
import ipx.ops as ops

pytestmark = pytest.mark.gpu

# (h, w): ragged widths, single row/column, sub-16-byte rows, several 1 KiB strips, strip-edge cases
SHAPES = [(1, 1), (1, 5), (7, 1), (2, 2), (3, 17), (16, 16), (5, 64), (33, 70), (9, 341), (40, 342), (21, 1030),
          (70, 683), (18, 2051), (130, 96)]


def rnd(shape, seed=0):
    return np.random.default_rng(seed).integers(0, 256, shape, dtype=np.uint8)


def dev_t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def host(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("cn", [3, 4])
def test_bgr2gray(dev, shape, cn):
    img = rnd(shape + (cn,), 1)
    np.testing.assert_array_equal(host(ops.bgr2gray(dev_t(img, dev))), O.bgr2gray_u8(img))


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("cn", [1, 3])
def test_lut_host_table(dev, shape, cn):
    img = rnd(shape + ((cn,) if cn > 1 else ()), 2)
    lut = np.random.default_rng(3).permutation(256).astype(np.uint8)
    np.testing.assert_array_equal(host(ops.apply_lut(dev_t(img, dev), lut)), lut[img])


def test_lut_device_tables_per_image(dev):
    imgs = rnd((5, 37, 53, 1), 4)
    tabs = np.stack([np.random.default_rng(i).permutation(256).astype(np.uint8) for i in range(5)])
    out = host(ops.apply_lut_dev(dev_t(imgs, dev), dev_t(tabs, dev)))
    for i in range(5):
        np.testing.assert_array_equal(out[i, :, :, 0], tabs[i][imgs[i, :, :, 0]])
    out = host(ops.apply_lut_dev(dev_t(imgs, dev), dev_t(tabs[2], dev)))
    np.testing.assert_array_equal(out, tabs[2][imgs])


@pytest.mark.parametrize("shape", SHAPES)
def test_bgr2gray_lut_fused(dev, shape):
    img = rnd(shape + (3,), 5)
    lut = luts.gamma(0.6)
    np.testing.assert_array_equal(host(ops.bgr2gray_lut(dev_t(img, dev), lut)), lut[O.bgr2gray_u8(img)])


@pytest.mark.parametrize("shape", SHAPES + [(600, 700)])
def test_hist_and_equalize(dev, shape):
    img = rnd(shape, 6)
    t = dev_t(img, dev)
    np.testing.assert_array_equal(host(ops.hist(t)).astype(np.uint32), O.hist_u8(img))
    out, h, lut = ops.equalize_hist(t, return_aux=True)
    np.testing.assert_array_equal(host(h)[0].astype(np.uint32), O.hist_u8(img))
    np.testing.assert_array_equal(host(lut)[0], O.equalize_lut_from_hist(O.hist_u8(img)))
    np.testing.assert_array_equal(host(out), O.equalize_hist_u8(img))


@pytest.mark.parametrize("shape", SHAPES + [(300, 411)])
@pytest.mark.parametrize("cn", [3, 4])
def test_bgr2gray_equalize_fused(dev, shape, cn):
    img = rnd(shape + (cn,), 21)
    gray, eq, h, lut = ops.bgr2gray_equalize(dev_t(img, dev), return_aux=True)
    g = O.bgr2gray_u8(img)
    np.testing.assert_array_equal(host(gray), g)
    np.testing.assert_array_equal(host(h)[0].astype(np.uint32), O.hist_u8(g))
    np.testing.assert_array_equal(host(eq), O.equalize_hist_u8(g))
    imgs = rnd((3,) + shape + (cn,), 22)
    gray, eq = ops.bgr2gray_equalize(dev_t(imgs, dev))
    for i in range(3):
        np.testing.assert_array_equal(host(eq)[i, :, :, 0], O.equalize_hist_u8(O.bgr2gray_u8(imgs[i])))


def test_hist_distributions(dev):
    """all-one-bin (LDS contention worst case), low contrast, two-level, missing low bins."""
    rng = np.random.default_rng(7)
    cases = [np.full((257, 300), 255, np.uint8), np.full((64, 64), 0, np.uint8),
             np.clip(rng.normal(128, 10, (300, 333)), 0, 255).astype(np.uint8),
             (rng.integers(0, 2, (100, 90)) * 200 + 17).astype(np.uint8),
             rng.integers(100, 256, (99, 101), dtype=np.uint8)]
    for img in cases:
        t = dev_t(img, dev)
        np.testing.assert_array_equal(host(ops.hist(t)).astype(np.uint32), O.hist_u8(img))
        np.testing.assert_array_equal(host(ops.equalize_hist(t)), O.equalize_hist_u8(img))


def test_equalize_batch_each_image_own_table(dev):
    imgs = np.stack([np.clip(np.random.default_rng(i).normal(60 + 30 * i, 8 + 5 * i, (45, 67)), 0, 255).astype(np.uint8)
                     for i in range(6)])[..., None]
    out = host(ops.equalize_hist(dev_t(imgs, dev)))
    for i in range(6):
        np.testing.assert_array_equal(out[i, :, :, 0], O.equalize_hist_u8(imgs[i, :, :, 0]))


def test_hist_ticketed_finish_one_workspace_over_many_geometries(dev):
    """For images cut into 65 or more blocks the last block to arrive sums the partial counts (two levels) and builds the
    table inside the counting launch: tickets at the head of the workspace, zeroed ONCE by ipx_hist_workspace_init and put
    back by the kernel.  One workspace is reused, without re-initialisation, over batches on both sides of that limit:
    1, 22, 32, 64 blocks per image (separate finish kernel), 65 (two full groups and one of a single block), 138 and 163
    (ragged last group), 1024 (32 x 32), several images with their own tickets, more images than tickets, then earlier
    geometries again.  All through the C ABI: ipx_hist_u8, ipx_equalize_hist_u8, ipx_bgr2gray_equalize_u8, and the
    NULL-workspace path."""
    import ipx._lib as L
    lib = L.load()
    rng = np.random.default_rng(91)
    geoms = [(1, 40, 50), (1, 700, 2048), (1, 2600, 4096), (3, 1024, 2048), (2, 3000, 3000), (1, 1024, 4096), (1, 1040, 4096),
             (24, 300, 1000), (5000, 6, 9), (1, 8200, 8192), (7, 1100, 4096), (1, 40, 50), (1, 2600, 4096)]
    big = max(lib.ipx_hist_workspace_bytes(*g) for g in geoms)
    ws = torch.full((big,), 0xA5, dtype=torch.uint8, device=dev)          # garbage everywhere ...
    L.call("ipx_hist_workspace_init", None, ws.data_ptr(), ws.numel())     # ... but the tickets
    for rep, (n, h, w) in enumerate(geoms):
        imgs = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        if rep % 3 == 1:
            imgs[0, : h // 2] = 7                                          # a heavy bin; i0 > 0 for some images
            imgs[0] = np.maximum(imgs[0], 7)
        t = dev_t(imgs, dev)
        want_h = np.stack([np.bincount(imgs[i].ravel(), minlength=256) for i in range(n)])
        want_l = np.stack([O.equalize_lut_from_hist(want_h[i]) for i in range(n)]) if n <= 64 else None
        hst = torch.zeros((n, 256), dtype=torch.int32, device=dev)
        L.call("ipx_hist_u8", None, t.data_ptr(), n, h, w, w, w * h, hst.data_ptr(), ws.data_ptr(), ws.numel())
        np.testing.assert_array_equal(host(hst), want_h, err_msg=f"hist {n}x{h}x{w}")
        out = torch.empty_like(t)
        hst.zero_()
        lut = torch.zeros((n, 256), dtype=torch.uint8, device=dev)
        L.call("ipx_equalize_hist_u8", None, t.data_ptr(), out.data_ptr(), n, h, w, w, w * h, w, w * h,
               hst.data_ptr(), lut.data_ptr(), ws.data_ptr(), ws.numel())
        np.testing.assert_array_equal(host(hst), want_h, err_msg=f"equalize hist {n}x{h}x{w}")
        if want_l is not None:
            np.testing.assert_array_equal(host(lut), want_l, err_msg=f"table {n}x{h}x{w}")
            got = host(out)
            for i in range(n):
                assert (got[i] == want_l[i][imgs[i]]).all(), f"pixels {n}x{h}x{w} image {i}"
        # the same call with no workspace (stream-ordered allocation, tickets cleared by the call) agrees
        out2, lut2 = torch.empty_like(t), torch.zeros_like(lut)
        L.call("ipx_equalize_hist_u8", None, t.data_ptr(), out2.data_ptr(), n, h, w, w, w * h, w, w * h,
               None, lut2.data_ptr(), None, 0)
        assert torch.equal(lut, lut2) and torch.equal(out, out2)
    # rows with a stride (units are cut per row), and the colour entry point on the same workspace
    n, h, w, pitch = 2, 700, 1500, 1536
    buf = rng.integers(0, 256, (n, h, pitch), dtype=np.uint8)
    t = dev_t(buf, dev)
    hst = torch.zeros((n, 256), dtype=torch.int32, device=dev)
    L.call("ipx_hist_u8", None, t.data_ptr(), n, h, w, pitch, pitch * h, hst.data_ptr(), ws.data_ptr(), ws.numel())
    np.testing.assert_array_equal(host(hst), np.stack([np.bincount(buf[i, :, :w].ravel(), minlength=256) for i in range(n)]))
    for shape in ((1, 2160, 3840, 3), (4, 500, 1200, 3), (1, 9, 11, 4)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        n, h, w, c = shape
        t = dev_t(img, dev)
        gray = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
        eq = torch.empty_like(gray)
        hst = torch.zeros((n, 256), dtype=torch.int32, device=dev)
        L.call("ipx_bgr2gray_equalize_u8", None, t.data_ptr(), gray.data_ptr(), eq.data_ptr(), n, h, w, c, w * c, w * c * h,
               w, w * h, w, w * h, hst.data_ptr(), None, ws.data_ptr(), ws.numel())
        for i in range(n):
            g = O.bgr2gray_u8(img[i])
            np.testing.assert_array_equal(host(hst)[i], np.bincount(g.ravel(), minlength=256))
            tab = O.equalize_lut_from_hist(np.bincount(g.ravel(), minlength=256))
            assert (host(eq)[i] == tab[g]).all()
    # the tickets are back at zero after all of that; everything behind them was scratch
    assert int(ws[: (4096 + 128) * 4].view(torch.int32).abs().sum()) == 0
    with pytest.raises(L.IpxError):
        L.call("ipx_hist_workspace_init", None, None, 1 << 20)
    with pytest.raises(L.IpxError):
        L.call("ipx_hist_workspace_init", None, ws.data_ptr(), 1024)


def test_equalize_lut_from_large_histograms(dev):
    """float32 scale / int->float rounding at 2^30 pixels: LUT kernel alone."""
    rng = np.random.default_rng(8)
    hs = []
    for i in range(16):
        p = rng.random(256) ** (1 + i % 4)
        if i % 3 == 0:
            p[: rng.integers(1, 200)] = 0
        h = np.floor(p / p.sum() * ((1 << 30) - 7 * i)).astype(np.int64)
        hs.append(h)
    hs.append(np.eye(256, dtype=np.int64)[37] * 12345)                # single-valued image
    hs.append(np.array([0] * 10 + [1] + [0] * 244 + [1 << 20], np.int64))
    H = np.stack(hs).astype(np.int32)
    got = host(ops.equalize_lut(dev_t(H, dev)))
    for i in range(len(hs)):
        np.testing.assert_array_equal(got[i], O.equalize_lut_from_hist(hs[i]), err_msg=f"hist {i}")


CONV = [("laplacian_sharpen", O.laplacian_sharpen_u8), ("box3", O.box3_u8), ("sobel", O.sobel_u8)]


def test_log_tables_built_on_the_device(dev):
    """pipelines.log_tables_dev == luts.log (the reference's numpy expressions) for every possible highest level present,
    several c; all-zero images included (level 0 only)."""
    import ipx.pipelines as P
    for c in (0.5, 1.0, 2.0, 7.3):
        hists = np.zeros((256, 256), np.int32)
        for top in range(256):
            hists[top, top] = 5
            if top > 3:
                hists[top, top // 2] = 1
        got = host(P.log_tables_dev(dev_t(hists, dev), c))
        for top in range(256):
            want = luts.log(c, hists[top])
            lv = np.nonzero(hists[top])[0]                        # only levels that occur are ever looked up
            np.testing.assert_array_equal(got[top][:top + 1], want[:top + 1], err_msg=f"c={c} top={top}")
            assert (got[top][lv] == want[lv]).all()


def test_log_lut_entry_point_strided_output_and_bad_arguments(dev):
    """ipx_log_lut_u8 through ops.log_lut: tables written with a row stride (one slot of N x k x 256 table sets) equal the
    contiguous ones and luts.log; the other slots stay untouched; bad arguments are rejected."""
    import ipx._lib as L
    rng = np.random.default_rng(5)
    hists = np.zeros((7, 256), np.int32)
    for i in range(7):
        lv = rng.integers(0, 256, rng.integers(1, 6))
        hists[i, lv] = rng.integers(1, 1000, lv.size)
    curve = torch.from_numpy(luts.log_curve(2.0)).to(dev)
    h = dev_t(hists, dev)
    plain = host(ops.log_lut(h, curve))
    sets = torch.full((7, 3, 256), 77, dtype=torch.uint8, device=dev)
    ops.log_lut(h, curve, out=sets[:, 1])
    got = host(sets)
    np.testing.assert_array_equal(got[:, 1], plain)
    assert (got[:, 0] == 77).all() and (got[:, 2] == 77).all()
    for i in range(7):
        top = int(np.nonzero(hists[i])[0].max())
        np.testing.assert_array_equal(plain[i][:top + 1], luts.log(2.0, hists[i])[:top + 1])
        assert (plain[i][top + 1:] == 255).all()
    with pytest.raises(ValueError):
        ops.log_lut(h, curve.double())
    with pytest.raises(ValueError):
        ops.log_lut(h, curve, out=torch.empty((7, 128), dtype=torch.uint8, device=dev))
    with pytest.raises(L.IpxError):
        L.call("ipx_log_lut_u8", 0, h.data_ptr(), 7, curve.data_ptr(), sets.data_ptr(), 100)
    with pytest.raises(L.IpxError):
        L.call("ipx_log_lut_u8", 0, h.data_ptr(), 0, curve.data_ptr(), sets.data_ptr(), 256)
    with pytest.raises(L.IpxError):
        L.call("ipx_log_lut_u8", 0, None, 7, curve.data_ptr(), sets.data_ptr(), 256)


def test_bgr2gray_hist_without_table_pass(dev):
    """ipx_bgr2gray_equalize_u8 with dst = NULL: gray image, histograms and equalize tables only (the caller applies the tables,
    e.g. in ipx_lut_multi_u8); rejected when no output would take the result."""
    import ipx._lib as L
    rng = np.random.default_rng(88)
    for shape in ((1, 5, 7, 3), (3, 40, 333, 3), (2, 64, 1100, 4)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        gray, hst, lut = ops.bgr2gray_hist(dev_t(img, dev))
        for i in range(shape[0]):
            want_gray = O.bgr2gray_u8(img[i])
            np.testing.assert_array_equal(host(gray)[i, :, :, 0], want_gray)
            np.testing.assert_array_equal(host(hst)[i], np.bincount(want_gray.ravel(), minlength=256))
            np.testing.assert_array_equal(host(lut)[i], O.equalize_lut_from_hist(np.bincount(want_gray.ravel(), minlength=256)))
    t = dev_t(rng.integers(0, 256, (1, 8, 8, 3), dtype=np.uint8), dev)
    g = torch.empty((1, 8, 8, 1), dtype=torch.uint8, device=dev)
    with pytest.raises(L.IpxError):
        L.call("ipx_bgr2gray_equalize_u8", None, t.data_ptr(), g.data_ptr(), None, 1, 8, 8, 3, 24, 192, 8, 64, 8, 64, None, None, None, 0)


def test_multi_table_pass(dev):
    """ipx_lut_multi_u8: k tables on one read of the source == k single table passes; shared and per-image tables, every k,
    ragged rows, colour, a strided batch with an unaligned base; argument errors."""
    import ipx._lib as L
    rng = np.random.default_rng(66)
    for shape in ((1, 1, 1, 1), (2, 7, 33, 1), (3, 40, 1111, 1), (2, 19, 300, 3), (1, 5, 16400, 1)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        t = dev_t(img, dev)
        for k in (1, 2, 5, 8):
            shared = rng.integers(0, 256, (k, 256), dtype=np.uint8)
            got = ops.apply_luts_dev(t, dev_t(shared, dev))
            assert len(got) == k
            for j in range(k):
                np.testing.assert_array_equal(host(got[j]), shared[j][img], err_msg=f"{shape} k={k} j={j}")
            per = rng.integers(0, 256, (shape[0], k, 256), dtype=np.uint8)
            got = ops.apply_luts_dev(t, dev_t(per, dev))
            for j in range(k):
                for i in range(shape[0]):
                    np.testing.assert_array_equal(host(got[j])[i], per[i, j][img[i]], err_msg=f"{shape} k={k} img={i} j={j}")
    big = rng.integers(0, 256, (3, 50, 1300), dtype=np.uint8)
    view = dev_t(big, dev)[:, 3:47, 5:5 + 1111].unsqueeze(-1)
    tabs = rng.integers(0, 256, (3, 4, 256), dtype=np.uint8)
    got = ops.apply_luts_dev(view, dev_t(tabs, dev))
    for j in range(4):
        for i in range(3):
            np.testing.assert_array_equal(host(got[j])[i, :, :, 0], tabs[i, j][big[i, 3:47, 5:1116]])
    t = dev_t(rng.integers(0, 256, (2, 8, 8, 1), dtype=np.uint8), dev)
    with pytest.raises(ValueError):
        ops.apply_luts_dev(t, dev_t(np.zeros((9, 256), np.uint8), dev))
    with pytest.raises(ValueError):
        ops.apply_luts_dev(t, dev_t(np.zeros((3, 2, 256), np.uint8), dev))       # 3 table sets for 2 images
    import ctypes
    tab = dev_t(np.zeros((2, 256), np.uint8), dev)
    out = torch.empty_like(t)
    ptrs = (ctypes.c_void_p * 2)(out.data_ptr(), out.data_ptr())                 # the same destination twice
    with pytest.raises(L.IpxError):
        L.call("ipx_lut_multi_u8", None, t.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p), 2, 2, 8, 8, 1, 8, 64, 8, 64, tab.data_ptr(), 0)
    out2 = torch.empty_like(t)
    ptrs = (ctypes.c_void_p * 2)(t.data_ptr(), out2.data_ptr())                  # a destination that is the source (k > 1)
    with pytest.raises(L.IpxError) as e:
        L.call("ipx_lut_multi_u8", None, t.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p), 2, 2, 8, 8, 1, 8, 64, 8, 64, tab.data_ptr(), 0)
    assert "overlaps the source" in str(e.value)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("cn", [1, 2, 3, 4])
@pytest.mark.parametrize("name,ref", CONV)
def test_conv3(dev, shape, cn, name, ref):
    img = rnd(shape + ((cn,) if cn > 1 else ()), 9)
    got = host(getattr(ops, name)(dev_t(img, dev)))
    np.testing.assert_array_equal(got, ref(img))


def test_gaussian_and_gray_in_one_read(dev):
    """ipx_gaussian_blur_gray_u8: the blurred colour batch and the gray image of the unblurred source from one read == the two
    entry points == the oracle.  Shapes that take the one-launch band-tile form (rows of whole 16-byte lanes >= 4 KiB: widths
    1366 (one tile, 4098 bytes -> not a lane multiple: two calls), 1376, 2048, 2752, 3840 px; heights around the 8-row
    tiles; pixel groups that straddle tile edges), shapes that fall back to the two calls, k = 3 / 5 / 7, a batch, a row-strided view."""
    rng = np.random.default_rng(321)
    for (h, w), k in (((9, 1376), 5), ((17, 2048), 5), ((8, 2752), 5), ((23, 3840), 5), ((16, 1376), 3), ((33, 4112), 3), ((7, 7680), 5),
                      ((20, 1366), 5), ((12, 300), 5), ((40, 2048), 7), ((5, 1), 3)):
        for n in (1, 2):
            img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
            g5, gray = ops.gaussian_blur_gray(dev_t(img, dev), k)
            for i in range(n):
                np.testing.assert_array_equal(host(g5)[i], O.gaussian_blur_u8(img[i], k), err_msg=f"gauss {n}x{h}x{w} k={k}")
                np.testing.assert_array_equal(host(gray)[i, :, :, 0], O.bgr2gray_u8(img[i]), err_msg=f"gray {n}x{h}x{w} k={k}")
    sat = np.zeros((16, 1376, 3), np.uint8); sat[..., 1] = 255; sat[:, ::2] = 255
    g5, gray = ops.gaussian_blur_gray(dev_t(sat, dev), 5)
    np.testing.assert_array_equal(host(gray), O.bgr2gray_u8(sat))
    np.testing.assert_array_equal(host(g5), O.gaussian_blur_u8(sat, 5))
    big = rng.integers(0, 256, (2, 40, 1500 * 3 + 48), dtype=np.uint8)
    view = dev_t(big, dev)[:, 3:37, 48:48 + 1376 * 3].unflatten(2, (1376, 3))
    ref = big[:, 3:37, 48:48 + 1376 * 3].reshape(2, 34, 1376, 3)
    g5, gray = ops.gaussian_blur_gray(view, 5)
    for i in range(2):
        np.testing.assert_array_equal(host(g5)[i], O.gaussian_blur_u8(np.ascontiguousarray(ref[i]), 5))
        np.testing.assert_array_equal(host(gray)[i, :, :, 0], O.bgr2gray_u8(np.ascontiguousarray(ref[i])))
    with pytest.raises(ValueError):
        ops.gaussian_blur_gray(dev_t(rng.integers(0, 256, (8, 64), dtype=np.uint8), dev), 5)


def test_conv3_and_table_pass_in_one_read_with_halo_rows(dev):
    """ipx_conv3_lut_u8: each 3x3 operator and a table pass of one gray batch from one read == the separate entry points == the
    oracle; with halo rows, a row block of a larger image gives the rows of the whole image's result (the row-partitioned
    configs[4] step: table pass + filter after the histogram all-reduce)."""
    rng = np.random.default_rng(325)
    refs = {"laplacian_sharpen": O.laplacian_sharpen_u8, "box3": O.box3_u8, "sobel": O.sobel_u8}
    for (n, h, w) in ((1, 1, 1), (2, 5, 43), (1, 33, 44), (3, 40, 333), (1, 70, 4096), (2, 19, 1000)):
        img = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        tabs = np.stack([rng.permutation(256).astype(np.uint8) for _ in range(n)])
        for name, ref in refs.items():
            f, tb = ops.conv3_lut(name, dev_t(img[..., None], dev), torch.from_numpy(tabs).to(dev))
            for i in range(n):
                np.testing.assert_array_equal(host(f)[i, :, :, 0], ref(img[i]), err_msg=f"{name} {n}x{h}x{w}")
                np.testing.assert_array_equal(host(tb)[i, :, :, 0], tabs[i][img[i]], err_msg=f"table {n}x{h}x{w}")
    whole = rng.integers(0, 256, (90, 700), dtype=np.uint8)
    tab = rng.permutation(256).astype(np.uint8)
    for y0, y1 in ((0, 30), (30, 61), (61, 90)):
        top = dev_t(whole[y0 - 1:y0], dev) if y0 > 0 else None
        bot = dev_t(whole[y1:y1 + 1], dev) if y1 < 90 else None
        for name, ref in refs.items():
            f, tb = ops.conv3_lut(name, dev_t(whole[y0:y1], dev), torch.from_numpy(tab).to(dev), top_halo=top, bottom_halo=bot)
            np.testing.assert_array_equal(host(f), ref(whole)[y0:y1], err_msg=f"{name} rows {y0}:{y1}")
            np.testing.assert_array_equal(host(tb), tab[whole[y0:y1]])
    with pytest.raises(ValueError):
        ops.conv3_lut("median", dev_t(whole, dev), torch.from_numpy(tab).to(dev))


def test_gaussian_sharpen_and_gray_in_one_read(dev):
    """ipx_gaussian_sharpen_gray_u8: the 5-tap Gaussian, the Laplacian sharpening and the gray image of one BGR batch from one read
    == the three entry points == the oracle: shapes that take the one-launch band-tile form (1080p- and 4K-wide rows, heights around
    the 8-row tiles, a last band of 1 ... 7 rows), shapes and kernel sizes that fall back, saturating pixels, a row-strided view."""
    rng = np.random.default_rng(324)
    for (h, w), k in (((9, 1920), 5), ((16, 1920), 5), ((21, 3840), 5), ((1, 2048), 5), ((2, 1376), 5), ((31, 1184), 5), ((12, 7680), 5),
                      ((20, 1366), 5), ((12, 300), 5), ((10, 1920), 3), ((9, 2048), 7)):
        for n in (1, 2):
            img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
            g5, sh, gray = ops.gaussian_sharpen_gray(dev_t(img, dev), k)
            for i in range(n):
                np.testing.assert_array_equal(host(g5)[i], O.gaussian_blur_u8(img[i], k), err_msg=f"gauss {n}x{h}x{w} k={k}")
                np.testing.assert_array_equal(host(sh)[i], O.laplacian_sharpen_u8(img[i]), err_msg=f"sharpen {n}x{h}x{w} k={k}")
                np.testing.assert_array_equal(host(gray)[i, :, :, 0], O.bgr2gray_u8(img[i]), err_msg=f"gray {n}x{h}x{w} k={k}")
    sat = np.zeros((19, 1920, 3), np.uint8); sat[::2, ::3] = 255; sat[5:9, 100:900, 1] = 255        # clip at both ends
    g5, sh, gray = ops.gaussian_sharpen_gray(dev_t(sat, dev), 5)
    np.testing.assert_array_equal(host(sh), O.laplacian_sharpen_u8(sat))
    np.testing.assert_array_equal(host(g5), O.gaussian_blur_u8(sat, 5))
    np.testing.assert_array_equal(host(gray), O.bgr2gray_u8(sat))
    big = rng.integers(0, 256, (2, 40, 2000 * 3 + 48), dtype=np.uint8)
    view = dev_t(big, dev)[:, 3:37, 48:48 + 1920 * 3].unflatten(2, (1920, 3))
    ref = big[:, 3:37, 48:48 + 1920 * 3].reshape(2, 34, 1920, 3)
    g5, sh, gray = ops.gaussian_sharpen_gray(view, 5)
    for i in range(2):
        np.testing.assert_array_equal(host(g5)[i], O.gaussian_blur_u8(np.ascontiguousarray(ref[i]), 5))
        np.testing.assert_array_equal(host(sh)[i], O.laplacian_sharpen_u8(np.ascontiguousarray(ref[i])))
        np.testing.assert_array_equal(host(gray)[i, :, :, 0], O.bgr2gray_u8(np.ascontiguousarray(ref[i])))


def test_neighbourhood_ops_refuse_overlap_but_take_two_halves_of_one_buffer(dev):
    """A destination that overlaps the source is refused by the neighbourhood operators (they would read pixels already
    overwritten); the left and right half of ONE pitched buffer interleave in memory without sharing a byte and are accepted."""
    import ipx._lib as L
    rng = np.random.default_rng(323)
    h, w = 37, 200
    buf = dev_t(rng.integers(0, 256, (h, 2 * w), dtype=np.uint8), dev)
    src = host(buf)[:, :w].copy()
    st = torch.cuda.current_stream().cuda_stream
    L.call("ipx_sobel_u8", st, buf.data_ptr(), buf.data_ptr() + w, 1, h, w, 1, 2 * w, 2 * w * h, 2 * w, 2 * w * h, None, None)
    L.call("ipx_median_blur_u8", st, buf.data_ptr(), buf.data_ptr() + w, 1, h, w // 2, 2 * w, 2 * w * h, 2 * w, 2 * w * h, 5)
    torch.cuda.synchronize()
    got = host(buf)
    np.testing.assert_array_equal(got[:, :w], src)                                     # the source half is untouched
    np.testing.assert_array_equal(got[:, w:w + w // 2], O.median_blur_u8(src[:, :w // 2], 5))
    np.testing.assert_array_equal(got[:, w + w // 2:], O.sobel_u8(src)[:, w // 2:])
    for shift in (0, 1, w - 1, 2 * w * 3):                                             # same bytes, one column, one pixel short of disjoint, rows below
        with pytest.raises(L.IpxError):
            L.call("ipx_box3_u8", st, buf.data_ptr(), buf.data_ptr() + shift, 1, h - 3, w, 1, 2 * w, 2 * w * h, 2 * w, 2 * w * h, None, None)
    with pytest.raises(L.IpxError):
        L.call("ipx_lee_filter_u8", st, buf.data_ptr(), buf.data_ptr() + 5, 1, h - 1, w, 2 * w, 2 * w * h, 2 * w, 2 * w * h, 5)


def test_hist_tables_without_table_pass(dev):
    """ipx_equalize_hist_u8 with dst = NULL (ops.hist_tables): histograms and equalize tables only, small images (separate
    finish) and images of 65+ blocks (finish inside the counting launch); rejected when no output would take the result."""
    import ipx._lib as L
    rng = np.random.default_rng(322)
    for shape in ((1, 5, 7), (3, 40, 333), (2, 1200, 2100)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        hst, lut = ops.hist_tables(dev_t(img[..., None], dev))
        for i in range(shape[0]):
            want = np.bincount(img[i].ravel(), minlength=256)
            np.testing.assert_array_equal(host(hst)[i], want)
            np.testing.assert_array_equal(host(lut)[i], O.equalize_lut_from_hist(want))
    t = dev_t(rng.integers(0, 256, (8, 8), dtype=np.uint8), dev)
    with pytest.raises(L.IpxError):
        L.call("ipx_equalize_hist_u8", 0, t.data_ptr(), None, 1, 8, 8, 8, 64, 0, 0, None, None, None, 0)


def test_sobel_and_table_pass_in_one_read(dev):
    """ipx_sobel_lut_u8: Sobel and a per-image table pass of one gray batch in one launch == the two separate entry points
    == the oracle; rows shorter than the walker's 44 bytes (per-byte blocks), ragged widths, 1 KiB strip edges, segment
    edges, a batch with per-image tables, a shared table, a row-strided view; bad arguments."""
    import ipx._lib as L
    rng = np.random.default_rng(123)
    for h, w in ((1, 1), (2, 3), (5, 43), (7, 44), (9, 45), (33, 1023), (40, 1024), (41, 1025), (130, 2051), (300, 700), (64, 4096)):
        for n in (1, 3):
            img = rng.integers(0, 256, (n, h, w, 1), dtype=np.uint8)
            tabs = np.stack([rng.permutation(256).astype(np.uint8) for _ in range(n)])
            sb, tb = ops.sobel_lut(dev_t(img, dev), dev_t(tabs, dev))
            for i in range(n):
                np.testing.assert_array_equal(host(sb)[i, :, :, 0], O.sobel_u8(img[i, :, :, 0]), err_msg=f"sobel {n}x{h}x{w}")
                np.testing.assert_array_equal(host(tb)[i, :, :, 0], tabs[i][img[i, :, :, 0]], err_msg=f"table {n}x{h}x{w}")
    img = rng.integers(0, 256, (2, 90, 1500), dtype=np.uint8)
    one = rng.permutation(256).astype(np.uint8)
    view = dev_t(img, dev)[:, 5:85, 7:1400].unsqueeze(-1)
    sb, tb = ops.sobel_lut(view, dev_t(one, dev))
    for i in range(2):
        crop = np.ascontiguousarray(img[i, 5:85, 7:1400])
        np.testing.assert_array_equal(host(sb)[i, :, :, 0], O.sobel_u8(crop))
        np.testing.assert_array_equal(host(tb)[i, :, :, 0], one[crop])
    g2 = dev_t(img[0], dev)
    s1, t1 = ops.sobel_lut(g2, dev_t(one, dev))
    np.testing.assert_array_equal(host(s1), host(ops.sobel(g2)))
    np.testing.assert_array_equal(host(t1), host(ops.apply_lut(g2, one)))
    with pytest.raises(ValueError):
        ops.sobel_lut(g2, dev_t(np.zeros((3, 256), np.uint8), dev))
    out = torch.empty_like(g2)
    with pytest.raises(L.IpxError):
        L.call("ipx_sobel_lut_u8", 0, g2.data_ptr(), out.data_ptr(), out.data_ptr(), 1, 90, 1500, 1500, 0, 1500, 0, dev_t(one, dev).data_ptr(), 0)
    with pytest.raises(L.IpxError):
        L.call("ipx_sobel_lut_u8", 0, g2.data_ptr(), out.data_ptr(), torch.empty_like(g2).data_ptr(), 1, 90, 1500, 1500, 0, 1500, 0, None, 0)


@pytest.mark.parametrize("name,ref", CONV)
def test_conv3_extremes(dev, name, ref):
    """saturating inputs: checkerboards of 0/255 hit both clamps of sharpen and the 255 cap of Sobel."""
    yy, xx = np.mgrid[0:50, 0:1100]
    for img in (((yy + xx) % 2 * 255).astype(np.uint8), ((yy // 3 + xx // 5) % 2 * 255).astype(np.uint8),
                np.full((50, 1100), 255, np.uint8)):
        for a in (img, np.stack([img, 255 - img, img], -1)):
            np.testing.assert_array_equal(host(getattr(ops, name)(dev_t(a, dev))), ref(a))


@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (7, 1), (33, 70), (40, 342), (18, 2051), (130, 96)])
@pytest.mark.parametrize("cn", [1, 3, 4])
@pytest.mark.parametrize("k", [1, 3, 5, 7, 9, 15, 31])
def test_gaussian(dev, shape, cn, k):
    img = rnd(shape + ((cn,) if cn > 1 else ()), 10)
    np.testing.assert_array_equal(host(ops.gaussian_blur(dev_t(img, dev), k)), O.gaussian_blur_u8(img, k))


@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_kernels_beyond_31_taps_two_pass_dot_product_path(dev, cn):
    """k > 31 (cv2.GaussianBlur(img, (g, g), 0) with the reference's unclamped `gauss`, server.py:402-409, :428): two passes through
    a Q8.8 scratch image on v_dot4_u32_u8 / v_dot2_u32_u16.  Images smaller than the radius (REFLECT_101 folds several times),
    widths around the 1 KiB row chunks and the 64-column tiles, heights around the 64-row tiles, every tap count parity case,
    a batch, a row-strided view, halo rows, saturated input; and taps that do not fit a byte (the plain two-pass kernels)."""
    rng = np.random.default_rng(40 + cn)
    for (h, w), k in (((1, 1), 33), ((3, 5), 35), ((20, 30), 63), ((65, 70), 33), ((64, 1100 // cn), 45), ((130, 257), 101),
                      ((40, 90), 255), ((129, 64), 127)):
        img = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
        np.testing.assert_array_equal(host(ops.gaussian_blur(dev_t(img, dev), k)), O.gaussian_blur_u8(img, k), err_msg=f"{h}x{w}x{cn} k={k}")
    full = np.full((70, 300, cn) if cn > 1 else (70, 300), 255, np.uint8)
    np.testing.assert_array_equal(host(ops.gaussian_blur(dev_t(full, dev), 63)), full)
    batch = rng.integers(0, 256, (3, 80, 333, cn), dtype=np.uint8)
    view = dev_t(batch, dev)[:, 4:76, 3:320]
    got = host(ops.gaussian_blur(view, 33))
    for i in range(3):
        ref = O.gaussian_blur_u8(np.ascontiguousarray(batch[i, 4:76, 3:320, 0] if cn == 1 else batch[i, 4:76, 3:320]), 33)
        np.testing.assert_array_equal(got[i].reshape(ref.shape), ref)
    # a row block with halo rows == the whole-image result cropped
    tall = rng.integers(0, 256, (150, 200, cn) if cn > 1 else (150, 200), dtype=np.uint8)
    t = dev_t(tall, dev)
    r = 16
    got = host(ops.gaussian_blur(t[40:100], 33, top_halo=t[40 - r:40], bottom_halo=t[100:100 + r]))
    np.testing.assert_array_equal(got, O.gaussian_blur_u8(tall, 33)[40:100])
    got = host(ops.gaussian_blur(t[:60], 33, bottom_halo=t[60:60 + r]))
    np.testing.assert_array_equal(got, O.gaussian_blur_u8(tall, 33)[:60])
    # taps above 255 cannot go through the byte dot product
    kx = [0] * 16 + [256] + [0] * 16
    ky = [1] * 16 + [224] + [1] * 16
    img = rng.integers(0, 256, (50, 120, cn) if cn > 1 else (50, 120), dtype=np.uint8)
    np.testing.assert_array_equal(host(ops.sep_filter_q8(dev_t(img, dev), kx, ky)), O.sep_filter_q8_u8(img, np.array(kx), np.array(ky)))
    np.testing.assert_array_equal(host(ops.sep_filter_q8(dev_t(img, dev), ky, ky[::-1])), O.sep_filter_q8_u8(img, np.array(ky), np.array(ky[::-1])))


def test_gaussian_c2_and_asymmetric_taps(dev):
    img = rnd((37, 300, 2), 11)
    np.testing.assert_array_equal(host(ops.gaussian_blur(dev_t(img, dev), 5)), O.gaussian_blur_u8(img, 5))
    img = rnd((37, 300, 3), 12)
    kx, ky = [200, 50, 6], [1, 2, 3, 4, 246][:3] + []      # 3 taps each; asymmetric on purpose
    ky = [10, 240, 6]
    np.testing.assert_array_equal(host(ops.sep_filter_q8(dev_t(img, dev), kx, ky)), O.sep_filter_q8_u8(img, kx, ky))
    kx5, ky5 = [1, 2, 3, 4, 246], [100, 50, 6, 60, 40]
    np.testing.assert_array_equal(host(ops.sep_filter_q8(dev_t(img, dev), kx5, ky5)), O.sep_filter_q8_u8(img, kx5, ky5))
    k9 = [1, 2, 3, 4, 236, 4, 3, 2, 1]
    np.testing.assert_array_equal(host(ops.sep_filter_q8(dev_t(img, dev), k9, k9[::-1])), O.sep_filter_q8_u8(img, k9, k9[::-1]))


@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_large_kernels_matrix_core_path(dev, cn):
    """9..31 taps run as banded int8 matrix products (ipx_sepmfma.hip): asymmetric taps (kx != ky, neither palindromic),
    every channel count, shapes around the 64 x 128 tile, batches with padded strides, extreme values."""
    rng = np.random.default_rng(40 + cn)
    def taps(k):
        t = rng.integers(0, 9, k)
        t[rng.integers(0, k)] += 256 - t.sum()              # sums to 256, every tap <= 127 for k >= 9? enforce below
        while t.max() > 127:
            i, j = int(t.argmax()), int(t.argmin())
            t[i] -= 20; t[j] += 20
        assert t.sum() == 256 and t.min() >= 0
        return [int(v) for v in t]
    for shape, k in (((70, 50), 9), ((64, 128), 13), ((65, 129), 17), ((131, 300), 31), ((9, 700), 21), ((200, 33), 27)):
        img = rng.integers(0, 256, shape + ((cn,) if cn > 1 else ()), dtype=np.uint8)
        kx, ky = taps(k), taps(k)
        np.testing.assert_array_equal(host(ops.sep_filter_q8(dev_t(img, dev), kx, ky)), O.sep_filter_q8_u8(img, kx, ky),
                                      err_msg=f"{shape} k={k}")
    ext = np.where(rng.random((90, 260, cn)) < 0.5, 0, 255).astype(np.uint8)
    ext = ext[:, :, 0] if cn == 1 else ext
    for k in (11, 31):
        np.testing.assert_array_equal(host(ops.gaussian_blur(dev_t(ext, dev), k)), O.gaussian_blur_u8(ext, k))
    big = rng.integers(0, 256, (3, 80, 300 * cn + 5), dtype=np.uint8)
    view = dev_t(big, dev)[:, 3:75, 2:2 + 290 * cn].unflatten(2, (290, cn))      # strided batch, unaligned base
    ref_in = big[:, 3:75, 2:2 + 290 * cn].reshape(3, 72, 290, cn)
    got = host(ops.gaussian_blur(view, 15))
    for i in range(3):
        want = O.gaussian_blur_u8(ref_in[i] if cn > 1 else ref_in[i][:, :, 0], 15)
        np.testing.assert_array_equal(got[i] if cn > 1 else got[i][:, :, 0], want)


@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_large_kernels_wave_per_strip_path(dev, cn):
    """inputs of 2 MB and more take ipx_sepwave.hip (one wave per 128-byte column strip): interior segments (LDS-DMA ring,
    reflect-aware at the top / bottom edge), frame pieces (first / last strips, ragged widths and heights), both column-pass
    lags (k <= 17 and above), asymmetric taps, extreme values, batches with padded strides, row blocks with halo rows."""
    rng = np.random.default_rng(60 + cn)
    def taps(k):
        t = rng.integers(0, 9, k)
        t[rng.integers(0, k)] += 256 - t.sum()
        while t.max() > 127:
            i, j = int(t.argmax()), int(t.argmin())
            t[i] -= 20; t[j] += 20
        assert t.sum() == 256 and t.min() >= 0
        return [int(v) for v in t]
    wpx = lambda nbytes: (nbytes + cn - 1) // cn                      # pixels per row for about nbytes row bytes
    # (h, row bytes): whole strips and tiles; ragged last strip (L % 128, L % 16); h % 16 != 0; few tall strips; one wide tile row band
    for (h, nb), k in (((1024, 2048), 9), ((1000, 2100), 17), ((1031, 2061), 19), ((4100, 520), 31), ((70, 30000), 13), ((2050, 1030), 25)):
        w = wpx(nb)
        img = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
        kx, ky = taps(k), taps(k)
        np.testing.assert_array_equal(host(ops.sep_filter_q8(dev_t(img, dev), kx, ky)), O.sep_filter_q8_u8(img, kx, ky),
                                      err_msg=f"{img.shape} k={k}")
    ext = np.where(rng.random((900, wpx(2400), cn)) < 0.5, 0, 255).astype(np.uint8)
    ext = ext[:, :, 0] if cn == 1 else ext
    for k in (11, 31):
        np.testing.assert_array_equal(host(ops.gaussian_blur(dev_t(ext, dev), k)), O.gaussian_blur_u8(ext, k))
    w = wpx(1500)
    big = rng.integers(0, 256, (3, 520, (w + 10) * cn + 5), dtype=np.uint8)
    view = dev_t(big, dev)[:, 3:515, 2:2 + w * cn].unflatten(2, (w, cn))         # strided batch, unaligned base
    ref_in = big[:, 3:515, 2:2 + w * cn].reshape(3, 512, w, cn)
    for k in (15, 21):
        got = host(ops.gaussian_blur(view, k))
        for i in range(3):
            want = O.gaussian_blur_u8(ref_in[i] if cn > 1 else ref_in[i][:, :, 0], k)
            np.testing.assert_array_equal(got[i] if cn > 1 else got[i][:, :, 0], want, err_msg=f"batch k={k}")
    img = rng.integers(0, 256, (1400, wpx(1800), cn) if cn > 1 else (1400, wpx(1800)), dtype=np.uint8)
    t = dev_t(img, dev)                                                         # row blocks with halo rows == the whole image cropped
    for k in (9, 31):
        r = k // 2
        want = O.gaussian_blur_u8(img, k)
        for y0, y1 in ((0, 1200), (100, 1400), (37, 1301)):
            top = t[y0 - r:y0] if y0 > 0 else None
            bot = t[y1:y1 + r] if y1 < 1400 else None
            got = host(ops.gaussian_blur(t[y0:y1], k, top_halo=top, bottom_halo=bot))
            np.testing.assert_array_equal(got, want[y0:y1], err_msg=f"rows {y0}:{y1} k={k}")


def test_gaussian_power_of_two_symmetric_taps(dev):
    """taps g * [1 a (b a) 1] with g a power of two (cv2's own 3- and 5-tap Gaussians among them) run in packed 16-bit
    arithmetic (sep_pow2_kernel); mixed scales per axis, the overflow boundary (which must fall back), 0/255 images."""
    rng = np.random.default_rng(77)
    cases = [([64, 128, 64], [64, 128, 64]), ([64, 128, 64], [32, 192, 32]), ([128, 0, 128], [64, 128, 64]),
             ([16, 64, 96, 64, 16], [16, 64, 96, 64, 16]), ([16, 64, 96, 64, 16], [32, 32, 128, 32, 32]),
             ([32, 64, 64, 64, 32], [64, 0, 128, 0, 64]),
             ([16, 64, 96, 64, 16], [8, 24, 192, 24, 8]),            # 255 * 16 * 32 does not fit 16 bits: general kernel
             ([1, 2, 250, 2, 1], [1, 2, 250, 2, 1])]
    for cn in (1, 2, 3, 4):
        for shape in ((40, 342), (9, 33), (130, 96), (3, 2051)):
            full = shape + ((cn,) if cn > 1 else ())
            imgs = [rng.integers(0, 256, full, dtype=np.uint8), np.where(rng.random(full) < 0.5, 0, 255).astype(np.uint8),
                    np.full(full, 255, np.uint8)]
            for kx, ky in cases:
                for img in imgs:
                    np.testing.assert_array_equal(host(ops.sep_filter_q8(dev_t(img, dev), kx, ky)),
                                                  O.sep_filter_q8_u8(img, kx, ky), err_msg=f"{full} {kx} {ky}")


def test_gaussian_band_tiles(dev):
    """rows that are a multiple of 16 bytes and at least 4 KiB long take the LDS band-tile kernel for cv2's 3- and 5-tap
    Gaussians: tile edges (a second tile of 2 lanes, exactly one tile, three tiles), heights around the 12-row bands,
    every channel count, 0 / 255 images, a strided batch with an unaligned base, and row blocks with halo rows."""
    rng = np.random.default_rng(55)
    for cn, w in ((1, 4096), (1, 4128), (2, 2064), (3, 1376), (4, 1032), (3, 3840), (1, 8192 + 16)):
        for h in (1, 2, 5, 12, 13, 23, 24, 25, 40):
            full = (h, w) + ((cn,) if cn > 1 else ())
            img = rng.integers(0, 256, full, dtype=np.uint8) if h != 13 else np.where(rng.random(full) < 0.5, 0, 255).astype(np.uint8)
            for k in (3, 5):
                np.testing.assert_array_equal(host(ops.gaussian_blur(dev_t(img, dev), k)), O.gaussian_blur_u8(img, k), err_msg=f"{full} k={k}")
            for name, ref in CONV:                                     # the 3 x 3 operators take band tiles on the same shapes
                np.testing.assert_array_equal(host(getattr(ops, name)(dev_t(img, dev))), ref(img), err_msg=f"{full} {name}")
    big = rng.integers(0, 256, (3, 40, 1400 * 3 + 9), dtype=np.uint8)
    view = dev_t(big, dev)[:, 3:38, 5:5 + 1376 * 3].unflatten(2, (1376, 3))      # strided batch, unaligned base, L = 4128
    ref_in = big[:, 3:38, 5:5 + 1376 * 3].reshape(3, 35, 1376, 3)
    for k in (3, 5):
        got = host(ops.gaussian_blur(view, k))
        for i in range(3):
            np.testing.assert_array_equal(got[i], O.gaussian_blur_u8(ref_in[i], k))
    for name, ref in CONV:
        got = host(getattr(ops, name)(view))
        for i in range(3):
            np.testing.assert_array_equal(got[i], ref(ref_in[i]), err_msg=name)
    img = rng.integers(0, 256, (60, 1376, 3), dtype=np.uint8)                   # a row block with halo rows == the whole image cropped
    t = dev_t(img, dev)
    for name, ref in CONV:
        want = ref(img)
        for y0, y1 in ((0, 17), (17, 43), (43, 60), (30, 31)):
            top = t[y0 - 1:y0] if y0 > 0 else None
            bot = t[y1:y1 + 1] if y1 < 60 else None
            np.testing.assert_array_equal(host(getattr(ops, name)(t[y0:y1], top_halo=top, bottom_halo=bot)), want[y0:y1], err_msg=f"{name} rows {y0}:{y1}")
    for k in (3, 5):
        r = k // 2
        want = O.gaussian_blur_u8(img, k)
        for y0, y1 in ((0, 17), (17, 43), (43, 60), (30, 31)):
            top = t[y0 - r:y0] if y0 > 0 else None
            bot = t[y1:y1 + r] if y1 < 60 else None
            got = host(ops.gaussian_blur(t[y0:y1], k, top_halo=top, bottom_halo=bot))
            np.testing.assert_array_equal(got, want[y0:y1], err_msg=f"rows {y0}:{y1} k={k}")


def test_gaussian_max_values_do_not_overflow(dev):
    img = np.full((40, 1200, 3), 255, np.uint8)
    for k in (3, 5, 7, 15, 31):
        np.testing.assert_array_equal(host(ops.gaussian_blur(dev_t(img, dev), k)), img)


def test_batches_and_strides(dev):
    """batch > 1, padded rows, padded image stride, unaligned base pointer."""
    big = rnd((3, 50, 400 * 3 + 7), 13)
    tb = dev_t(big, dev)
    view = tb[:, 5:45, 1:1 + 390 * 3].unflatten(2, (390, 3))          # N x 40 x 390 x 3, strided, base + 1
    ref_in = big[:, 5:45, 1:1 + 390 * 3].reshape(3, 40, 390, 3)
    for name, ref in CONV + [("g5", lambda a: O.gaussian_blur_u8(a, 5))]:
        f = (lambda t: ops.gaussian_blur(t, 5)) if name == "g5" else getattr(ops, name)
        got = host(f(view))
        for i in range(3):
            np.testing.assert_array_equal(got[i], ref(ref_in[i]), err_msg=name)
    got = host(ops.bgr2gray(view))
    for i in range(3):
        np.testing.assert_array_equal(got[i, :, :, 0], O.bgr2gray_u8(ref_in[i]))
    gview = tb[:, 3:47, 3:3 + 1111].unsqueeze(-1)
    gref = big[:, 3:47, 3:3 + 1111]
    lut = luts.negative()
    np.testing.assert_array_equal(host(ops.apply_lut(gview, lut))[..., 0], lut[gref])
    got = host(ops.equalize_hist(gview))
    for i in range(3):
        np.testing.assert_array_equal(got[i, :, :, 0], O.equalize_hist_u8(np.ascontiguousarray(gref[i])))


@pytest.mark.parametrize("cn", [1, 3])
def test_halo_rows_equal_whole_image(dev, cn):
    """A row shard filtered with exchanged halo rows == the same rows of the whole image filtered at once."""
    img = rnd((90, 500) + ((cn,) if cn > 1 else ()), 14)
    t = dev_t(img, dev)
    for name, ref, r in [("laplacian_sharpen", O.laplacian_sharpen_u8, 1), ("box3", O.box3_u8, 1),
                         ("sobel", O.sobel_u8, 1), ("g5", lambda a: O.gaussian_blur_u8(a, 5), 2),
                         ("g9", lambda a: O.gaussian_blur_u8(a, 9), 4), ("g31", lambda a: O.gaussian_blur_u8(a, 31), 15)]:
        whole = ref(img)
        for (y0, y1) in ((0, 30), (30, 61), (61, 90)):
            top = t[y0 - r:y0] if y0 > 0 else None
            bot = t[y1:y1 + r] if y1 < 90 else None
            if name.startswith("g"):
                got = ops.gaussian_blur(t[y0:y1], int(name[1:]), top_halo=top, bottom_halo=bot)
            else:
                got = getattr(ops, name)(t[y0:y1], top_halo=top, bottom_halo=bot)
            np.testing.assert_array_equal(host(got), whole[y0:y1], err_msg=f"{name} rows {y0}:{y1}")


@pytest.mark.parametrize("cn", [1, 3])
def test_halo_rows_block_shorter_than_radius_at_image_border(dev, cn):
    """A first / last row block of fewer rows than the radius: REFLECT_101 at the image border reaches past the block
    into the neighbour's halo rows (row -1 of a 1-row first block is global row 1 = the first bottom halo row)."""
    img = rnd((40, 300) + ((cn,) if cn > 1 else ()), 15)
    t = dev_t(img, dev)
    for name, ref, r in [("sobel", O.sobel_u8, 1), ("box3", O.box3_u8, 1), ("laplacian_sharpen", O.laplacian_sharpen_u8, 1),
                         ("g5", lambda a: O.gaussian_blur_u8(a, 5), 2), ("g7", lambda a: O.gaussian_blur_u8(a, 7), 3),
                         ("g15", lambda a: O.gaussian_blur_u8(a, 15), 7), ("g31", lambda a: O.gaussian_blur_u8(a, 31), 15)]:
        whole = ref(img)
        for rows in {1, max(1, r - 1), r}:
            for (y0, y1) in ((0, rows), (40 - rows, 40)):
                top = t[y0 - r:y0] if y0 > 0 else None
                bot = t[y1:y1 + r] if y1 < 40 else None
                if name.startswith("g"):
                    got = ops.gaussian_blur(t[y0:y1], int(name[1:]), top_halo=top, bottom_halo=bot)
                else:
                    got = getattr(ops, name)(t[y0:y1], top_halo=top, bottom_halo=bot)
                np.testing.assert_array_equal(host(got), whole[y0:y1], err_msg=f"{name} rows {y0}:{y1}")


def test_golden_server_cases_on_device(dev, golden):
    """The real server.py outputs (tests/golden) reproduced by host-built table + device apply."""
    g, meta = golden
    from test_oracle_golden import _lut_for
    n = 0
    for c in meta["cases"]:
        if c["status"] != 200:
            continue
        img, out = g["in_" + c["input"]], g["out_" + c["name"]]
        t = dev_t(img, dev)
        mode = c["mode"] or "gray8"
        if mode == "denoise_manual":                          # real lee_filter output through the real dispatch
            from test_oracle_golden import _lee_kernel
            np.testing.assert_array_equal(host(ops.lee_filter(t, _lee_kernel(c["form"]))), out, err_msg=c["name"])
            n += 1
            continue
        if mode in ("log", "logarithmic", "logrithmic"):
            present = host(ops.hist(t))                       # device histogram feeds s.max()
            lut = luts.log(float(c["form"].get("c", "2.0")), present)
        else:
            lut = _lut_for(mode, c["form"], img)
        np.testing.assert_array_equal(host(ops.apply_lut(t, lut)), out, err_msg=c["name"])
        n += 1
    assert n >= 60


@pytest.mark.parametrize("shape", [(1, 1), (5, 40), (28, 36), (33, 70), (64, 97), (130, 96)])
@pytest.mark.parametrize("wnd", [1, 3, 4, 5, 7, 9, 15, 31])
def test_lee_filter(dev, shape, wnd):
    """server.py:711-732 bit for bit (float32 mean / pairwise-summed variance per window), incl. images smaller than
    the window (all border) and flat patches (var == 0)."""
    rng = np.random.default_rng(shape[0] * 131 + wnd)
    img = np.clip(rng.normal(120, 40, shape), 0, 255).astype(np.uint8)
    if shape[0] > 12 and shape[1] > 20:
        img[3:12, 4:20] = 61
    np.testing.assert_array_equal(host(ops.lee_filter(dev_t(img, dev), wnd)), O.lee_filter_u8(img, wnd))


def test_lee_filter_golden_and_batch(dev, golden):
    g, _ = golden
    for w in (3, 5, 7):
        np.testing.assert_array_equal(host(ops.lee_filter(dev_t(g["in_lee"], dev), w)), g[f"out_lee_{w}"])
    for w in (9, 15, 31):
        np.testing.assert_array_equal(host(ops.lee_filter(dev_t(g["in_rand"], dev), w)), g[f"out_lee_rand_{w}"])
    # 3 x 3 windows the real server answers differently from an implementation that fuses a multiply into an add
    np.testing.assert_array_equal(host(ops.lee_filter(dev_t(g["in_lee_rounding"], dev), 3)), g["out_lee_rounding_3"])
    rng = np.random.default_rng(3)
    batch = rng.integers(0, 256, (3, 50, 77, 1), dtype=np.uint8)
    got = host(ops.lee_filter(dev_t(batch, dev), 5))
    for i in range(3):
        np.testing.assert_array_equal(got[i, :, :, 0], O.lee_filter_u8(batch[i, :, :, 0], 5))
    extremes = np.where(rng.random((40, 60)) < 0.5, 0, 255).astype(np.uint8)
    np.testing.assert_array_equal(host(ops.lee_filter(dev_t(extremes, dev), 7)), O.lee_filter_u8(extremes, 7))


def test_lee_wide_windows_two_pixels_per_thread(dev):
    """windows 11 ... 31 run on lee_wide_kernel (a thread owns two adjacent pixels, packed float32, numpy's pairwise order):
    every window size, widths around the 64-pixel tiles and odd widths (the last pair's right pixel is outside the image),
    heights around the 8-row tiles, a batch, a row-strided view, flat patches (var == 0) and extremes; golden fixtures
    of the real lee_filter for 13 ... 31 are in test_lee_filter_golden_and_batch."""
    rng = np.random.default_rng(77)
    for wnd in (10, 11, 12, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31):
        k = wnd // 2
        for h, w in ((2 * k + 1, 2 * k + 1), (2 * k + 2, 2 * k + 3), (40, 63), (41, 64), (39, 65), (33, 129), (48, 200)):
            img = np.clip(rng.normal(120, 45, (h, w)), 0, 255).astype(np.uint8)
            if h > 36 and w > 60:
                img[2:36, 10:60] = 200                                   # flat: var == 0 inside for the smaller windows
            np.testing.assert_array_equal(host(ops.lee_filter(dev_t(img, dev), wnd)), O.lee_filter_u8(img, wnd), err_msg=f"{h}x{w} w={wnd}")
    ext = (rng.integers(0, 2, (45, 90)) * 255).astype(np.uint8)
    for wnd in (13, 31):
        np.testing.assert_array_equal(host(ops.lee_filter(dev_t(ext, dev), wnd)), O.lee_filter_u8(ext, wnd))
    big = rng.integers(0, 256, (3, 50, 150), dtype=np.uint8)
    view = dev_t(big, dev)[:, 3:47, 5:136].unsqueeze(-1)
    got = host(ops.lee_filter(view, 15))
    for i in range(3):
        np.testing.assert_array_equal(got[i, :, :, 0], O.lee_filter_u8(np.ascontiguousarray(big[i, 3:47, 5:136]), 15))


def test_lee_column_walker(dev):
    """windows 3, 4 (= 5), 5, ... 9 on images at least 64 columns wide run on the column walker (lee_small_kernel; 10 and 11 take
    lee_wide_kernel since round 3 and are checked here on the same shapes): widths around
    the 64-column strips and 256-column blocks, heights around the row segments, flat patches (var == 0), two-level and
    near-constant images (tiny variances), a strided batch with an unaligned base; every pixel equal to numpy's result."""
    rng = np.random.default_rng(91)
    for h, w in ((1, 64), (2, 65), (3, 64), (5, 66), (6, 127), (64, 128), (65, 129), (130, 257), (200, 300), (7, 1031)):
        for wnd in (3, 4, 5, 6, 7, 8, 9, 10, 11):
            img = np.clip(rng.normal(120, 40, (h, w)), 0, 255).astype(np.uint8)
            if h > 12:
                img[3:12, 4:40] = 61
            np.testing.assert_array_equal(host(ops.lee_filter(dev_t(img, dev), wnd)), O.lee_filter_u8(img, wnd), err_msg=f"{h}x{w} w={wnd}")
    for img in (np.where(rng.random((90, 200)) < 0.5, 0, 255).astype(np.uint8),
                (100 + (rng.random((90, 200)) < 0.02)).astype(np.uint8), np.zeros((20, 70), np.uint8), np.full((20, 70), 255, np.uint8),
                rng.integers(0, 256, (300, 333), dtype=np.uint8)):
        for wnd in (3, 5, 7, 9, 11):
            np.testing.assert_array_equal(host(ops.lee_filter(dev_t(img, dev), wnd)), O.lee_filter_u8(img, wnd))
    big = rng.integers(0, 256, (3, 70, 400), dtype=np.uint8)
    view = dev_t(big, dev)[:, 5:65, 3:3 + 311].unsqueeze(-1)
    for wnd in (3, 5, 7, 9, 11):
        got = host(ops.lee_filter(view, wnd))
        for i in range(3):
            np.testing.assert_array_equal(got[i, :, :, 0], O.lee_filter_u8(np.ascontiguousarray(big[i, 5:65, 3:314]), wnd))


@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (5, 40), (33, 70), (64, 97), (130, 96)])
@pytest.mark.parametrize("k", [1, 3, 5, 7, 11, 31])
def test_median_blur(dev, shape, k):
    """cv2.medianBlur restated (BORDER_REPLICATE, exact median), incl. images smaller than the window."""
    rng = np.random.default_rng(shape[1] * 7 + k)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    if shape[0] > 8:
        img[rng.random(shape) < 0.1] = 255
        img[2:6, 3:9] = 17
    np.testing.assert_array_equal(host(ops.median_blur(dev_t(img, dev), k)), O.median_blur_u8(img, k))


def test_median_extremes_and_batch(dev):
    rng = np.random.default_rng(2)
    for img in (np.zeros((20, 30), np.uint8), np.full((20, 30), 255, np.uint8),
                np.where(rng.random((40, 50)) < 0.5, 0, 255).astype(np.uint8)):
        for k in (3, 9):
            np.testing.assert_array_equal(host(ops.median_blur(dev_t(img, dev), k)), O.median_blur_u8(img, k))
    batch = rng.integers(0, 256, (3, 37, 45, 1), dtype=np.uint8)
    got = host(ops.median_blur(dev_t(batch, dev), 5))
    for i in range(3):
        np.testing.assert_array_equal(got[i, :, :, 0], O.median_blur_u8(batch[i, :, :, 0], 5))


@pytest.mark.parametrize("k", [3, 5, 7, 9, 11])
def test_median_strip_walkers(dev, k):
    """k = 3, 5, 7, 9 and 11 on rows long enough for the strip walkers (sorted columns; shared pair / quad merges from k = 5): widths
    around the 16-byte lane and 1 KiB strip boundaries, heights around the row segments and below the window, few grey
    levels (many ties), ramps and their mirror images (every rank order of a window's columns), a strided batch with an
    unaligned base."""
    rng = np.random.default_rng(33 + k)
    for h, w in ((1, 36), (2, 47), (3, 48), (4, 40), (7, 49), (25, 64), (26, 1023), (50, 1024), (49, 1041), (130, 2051), (5, 4099)):
        for levels in (256, 3):
            img = (rng.integers(0, levels, (h, w)) * (255 // (levels - 1))).astype(np.uint8)
            np.testing.assert_array_equal(host(ops.median_blur(dev_t(img, dev), k)), O.median_blur_u8(img, k), err_msg=f"{h}x{w}")
    yy, xx = np.mgrid[0:70, 0:1100]
    for img in ((xx * 3 + yy * 5) % 256, (xx * 7 - yy * 11) % 256, (xx * yy) % 251, np.where((xx // 3 + yy // 2) % 2, 255, 0)):
        img = img.astype(np.uint8)
        for v in (img, img[::-1].copy(), img[:, ::-1].copy()):
            np.testing.assert_array_equal(host(ops.median_blur(dev_t(v, dev), k)), O.median_blur_u8(v, k))
    big = rng.integers(0, 256, (3, 60, 700), dtype=np.uint8)
    view = dev_t(big, dev)[:, 5:55, 3:3 + 611].unsqueeze(-1)
    got = host(ops.median_blur(view, k))
    for i in range(3):
        np.testing.assert_array_equal(got[i, :, :, 0], O.median_blur_u8(np.ascontiguousarray(big[i, 5:55, 3:614]), k))


@pytest.mark.parametrize("k", [9, 11, 13, 21, 31])
def test_median_many_strips_and_bands(dev, k):
    """Several 64-column strips and row bands per image, a strided batch, flat regions (all lanes bump one bin)."""
    rng = np.random.default_rng(k)
    batch = rng.integers(0, 256, (2, 300, 333, 1), dtype=np.uint8)
    batch[0, 40:120, 50:300] = 200                                   # flat patch wider than a strip
    batch[1, :, :, 0] = (batch[1, :, :, 0] // 64) * 64                # four grey levels only
    got = host(ops.median_blur(dev_t(batch, dev), k))
    for i in range(2):
        np.testing.assert_array_equal(got[i, :, :, 0], O.median_blur_u8(batch[i, :, :, 0], k))
    wide = rng.integers(0, 256, (40, 2100), dtype=np.uint8)          # one band, 33 strips
    np.testing.assert_array_equal(host(ops.median_blur(dev_t(wide, dev), k)), O.median_blur_u8(wide, k))


@pytest.mark.parametrize("k", list(range(13, 32, 2)))
def test_median_per_row_window_histograms_every_k(dev, k):
    """13..31 taps (one kernel instance each, ipx_medianrow.hip): a lane per row, packed 8-bit (13, 15 taps) / 16-bit counters.
    Constant images (one counter takes all k*k samples: 225 of 255, 961 of 65535), two levels split near the rank (the
    answer flips between parts of the packed dwords), values on both sides of every part boundary (63|64, 127|128,
    191|192), several 64-row bands with a ragged last one, several column segments, images smaller than the window and
    narrower than one staging chunk, a strided batch with an unaligned base."""
    rng = np.random.default_rng(100 + k)
    cases = [np.full((70, 140), 255, np.uint8), np.full((66, 40), 0, np.uint8), np.full((5, 7), 191, np.uint8),
             np.where(rng.random((90, 300)) < 0.5, 0, 255).astype(np.uint8),
             rng.choice(np.array([63, 64, 127, 128, 191, 192], np.uint8), (130, 200)),
             rng.integers(0, 256, (131, 517), dtype=np.uint8), rng.integers(0, 256, (3, 300), dtype=np.uint8),
             rng.integers(0, 256, (200, 9), dtype=np.uint8), (rng.integers(0, 4, (64, 256)) * 85).astype(np.uint8)]
    yy, xx = np.mgrid[0:100, 0:280]
    cases.append(((xx * 5 + yy * 3) % 256).astype(np.uint8))
    cases.append(np.where(xx < 140, 10, 240).astype(np.uint8))          # a vertical edge: every row's median jumps at the same x
    for img in cases:
        np.testing.assert_array_equal(host(ops.median_blur(dev_t(img, dev), k)), O.median_blur_u8(img, k), err_msg=str(img.shape))
    big = rng.integers(0, 256, (3, 90, 420), dtype=np.uint8)
    view = dev_t(big, dev)[:, 3:80, 5:5 + 403].unsqueeze(-1)
    got = host(ops.median_blur(view, k))
    for i in range(3):
        np.testing.assert_array_equal(got[i, :, :, 0], O.median_blur_u8(np.ascontiguousarray(big[i, 3:80, 5:408]), k))


@pytest.mark.parametrize("shape", [(1, 1), (4, 9), (33, 70), (64, 97)])
@pytest.mark.parametrize("m", [1, 3, 5, 7, 15, 31])
def test_morphology(dev, shape, m):
    """cv2.erode / dilate / morphologyEx(OPEN, CLOSE) restated with OpenCV's elliptical structuring element."""
    rng = np.random.default_rng(shape[1] + m)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    se = ops.ellipse_se(m)
    np.testing.assert_array_equal(se, O.ellipse_se(m))
    t = dev_t(img, dev)
    np.testing.assert_array_equal(host(ops.erode(t, se)), O.erode_u8(img, se))
    np.testing.assert_array_equal(host(ops.dilate(t, se)), O.dilate_u8(img, se))
    np.testing.assert_array_equal(host(ops.morph_open(t, se)), O.morph_open_u8(img, se))
    np.testing.assert_array_equal(host(ops.morph_close(t, se)), O.morph_close_u8(img, se))
    rect = np.ones((3, 5), np.uint8)                                  # non-square, asymmetric support
    rect[0, 0] = 0
    np.testing.assert_array_equal(host(ops.erode(t, rect)), O.erode_u8(img, rect))
    np.testing.assert_array_equal(host(ops.dilate(t, rect)), O.dilate_u8(img, rect))


def test_denoise_stats_on_device(dev):
    """server.py:190-211 bit for bit: numpy's float32 mean / var / std (order-dependent beyond 2^24) from the GPU sums."""
    rng = np.random.default_rng(9)
    for img in (rng.integers(0, 256, (300, 411), dtype=np.uint8),
                np.clip(rng.normal(128, 12, (256, 256)), 0, 255).astype(np.uint8),
                np.where(rng.random((200, 200)) < 0.05, 255, 100).astype(np.uint8),
                rng.integers(0, 256, (1, 5), dtype=np.uint8),
                rng.integers(0, 256, (64, 128), dtype=np.uint8),                 # exactly one 8192-element buffer
                rng.integers(200, 256, (2160, 3840), dtype=np.uint8),            # partial sums far beyond 2^24
                np.clip(rng.normal(40, 30, (1080, 1921)), 0, 255).astype(np.uint8)):
        ref = O.denoise_stats(img, "nlm")
        got = ops.denoise_stats(dev_t(img, dev), "nlm")
        assert got == ref, (img.shape, got, ref)
        lv = np.arange(256, dtype=np.float32)
        assert ops.pairwise_sum_lut(dev_t(img, dev), lv) == np.add.reduce(img.flatten().astype(np.float32))
    strided = dev_t(rng.integers(0, 256, (100, 300), dtype=np.uint8), dev)[:, 10:250]       # non-contiguous view
    assert ops.denoise_stats(strided, "lee") == O.denoise_stats(strided.cpu().numpy(), "lee")


def test_full_size_properties(dev):
    """BASELINE configs at full size through size-independent properties (no oracle pass at 4K x 32)."""
    n, h, w = 4, 2160, 3840
    g = torch.Generator(device="cpu").manual_seed(0)
    bgr = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, generator=g).to(dev)
    gray = ops.bgr2gray(bgr)
    hs = ops.hist(gray)
    assert hs.shape == (n, 256) and bool((hs.sum(1) == h * w).all())         # checksum of the histogram
    # histogram agrees with torch's own count
    for i in range(n):
        assert torch.equal(hs[i].to(torch.int64), torch.bincount(gray[i].flatten().to(torch.int64), minlength=256))
    eq = ops.equalize_hist(gray)
    # equalization is a monotone map of the gray levels: order of pixels is preserved
    a, b = gray[0, 100:200, 100:900, 0].flatten().to(torch.int32), eq[0, 100:200, 100:900, 0].flatten().to(torch.int32)
    o = torch.argsort(a, stable=True)
    assert bool((b[o][1:] >= b[o][:-1]).all())
    # negative twice is the identity; LUT linearity of composition
    neg = ops.apply_lut(gray, luts.negative())
    assert torch.equal(ops.apply_lut(neg, luts.negative()), gray)
    # constant image is a fixed point of every convolution; batch entries are independent
    const = torch.full((2, h, w, 3), 93, dtype=torch.uint8, device=dev)
    for f in (ops.laplacian_sharpen, ops.box3, lambda t: ops.gaussian_blur(t, 5), lambda t: ops.gaussian_blur(t, 15)):
        assert torch.equal(f(const), const)
    assert int(ops.sobel(const).max()) == 0
    s_all = ops.laplacian_sharpen(bgr)
    assert torch.equal(s_all[2], ops.laplacian_sharpen(bgr[2]))
    # one 4K image against the oracle on a band of rows (oracle on the band + 1 halo row each side)
    band = bgr[1, 999:1101].cpu().numpy()
    np.testing.assert_array_equal(s_all[1, 1000:1100].cpu().numpy(), O.laplacian_sharpen_u8(band)[1:-1])
    g5 = ops.gaussian_blur(bgr, 5)
    band = bgr[3, 0:66].cpu().numpy()
    np.testing.assert_array_equal(g5[3, 0:64].cpu().numpy(), O.gaussian_blur_u8(band, 5)[:64])
    sb = ops.sobel(gray)
    band = gray[0, 2060:2160, :, 0].cpu().numpy()
    np.testing.assert_array_equal(sb[0, 2061:2160, :, 0].cpu().numpy(), O.sobel_u8(band)[1:])


def test_full_size_rank_and_lee_filters(dev):
    """Median, morphology and the Lee filter on a 4K frame: random interior crops and the image corners against the
    oracle (crop + margin), plus order properties that need no oracle (erode <= src <= dilate, open <= src <= close,
    median of a constant / of a two-level image stays inside the levels)."""
    h, w = 2160, 3840
    rng = np.random.default_rng(77)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[500:900, 1000:2600] = (img[500:900, 1000:2600] // 128) * 255          # a two-level region: many equal values
    t = dev_t(img, dev)

    def crops(r, size=80):
        cs = [(int(rng.integers(r, h - size - r)), int(rng.integers(r, w - size - r))) for _ in range(4)]
        cs += [(600, 1500), (0, 0), (0, w - size), (h - size, 0), (h - size, w - size), (2100, 2000)]
        return cs, size

    def check(got, fn, r, replicate):
        cs, size = crops(r)
        for (y, x) in cs:
            y0, y1, x0, x1 = max(0, y - r), min(h, y + size + r), max(0, x - r), min(w, x + size + r)
            at_border = y0 == 0 or x0 == 0 or y1 == h or x1 == w
            ref = fn(img[y0:y1, x0:x1])                     # the crop's own border handling is only right AT the image border
            sub = ref[y - y0:y - y0 + size, x - x0:x - x0 + size]
            if at_border and not replicate:
                continue
            np.testing.assert_array_equal(got[y:y + size, x:x + size], sub, err_msg=f"crop {y},{x} r={r}")

    for k in (3, 5, 9, 11, 31):
        check(host(ops.median_blur(t, k)), lambda a: O.median_blur_u8(a, k), k // 2, True)
    hd = np.clip(rng.normal(110, 50, (1080, 1920)), 0, 255).astype(np.uint8)   # every pixel of a 1080p frame: 2 M windows each
    hd[300:500, 200:1500] = (hd[300:500, 200:1500] // 64) * 64               # few levels: tiny and zero variances
    for wnd in (3, 5, 7):
        np.testing.assert_array_equal(host(ops.lee_filter(dev_t(hd, dev), wnd)), O.lee_filter_u8(hd, wnd), err_msg=f"lee {wnd}")
    for wnd in (3, 5, 31):                                  # Lee copies the border band: only interior crops are comparable
        got = host(ops.lee_filter(t, wnd))
        check(got, lambda a: O.lee_filter_u8(a, wnd), wnd // 2 + wnd // 2, False)
        np.testing.assert_array_equal(got[:wnd // 2], img[:wnd // 2])
    se = ops.ellipse_se(7)
    er, di = ops.erode(t, se), ops.dilate(t, se)
    assert bool((er <= t).all()) and bool((t <= di).all())
    op, cl = ops.morph_open(t, se), ops.morph_close(t, se)
    assert bool((op <= t).all()) and bool((t <= cl).all())
    check(host(op), lambda a: O.morph_open_u8(a, se), 6, True)
    for msz in (15, 31):                                    # the run / sparse-table kernel at full size
        seb = ops.ellipse_se(msz)
        erb, dib = ops.erode(t, seb), ops.dilate(t, seb)
        assert bool((erb <= er).all()) and bool((di <= dib).all())          # a larger ellipse contains the smaller one
        check(host(erb), lambda a: O.erode_u8(a, seb), msz // 2, True)
        check(host(dib), lambda a: O.dilate_u8(a, seb), msz // 2, True)
    for wnd in (11, 15, 21):                                # the two-pixels-per-thread Lee kernel at full size
        got = host(ops.lee_filter(t, wnd))
        check(got, lambda a: O.lee_filter_u8(a, wnd), wnd // 2 + wnd // 2, False)
    for k in (33, 63):                                      # the dot-product two-pass filters at full size (interior crops: REFLECT_101
        got = host(ops.gaussian_blur(t, k))                 # at a crop's edge is only right at the image border)
        check(got, lambda a: O.gaussian_blur_u8(a, k), k // 2, False)
    two = dev_t(img[500:900, 1000:2600], dev)
    m = ops.median_blur(two, 7)
    assert set(torch.unique(m).tolist()) <= {0, 255}


def test_streaming_variants_of_the_kernels_in_a_child_process(dev):
    """The nt (streaming) template variants are chosen by working-set size, far above what tests use; IPX_NT_BYTES=0
    forces them (read once per process, hence the child process): same pixels."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = r'''
import sys, numpy as np, torch
sys.path[:0] = [%r, %r, %r]
import oracle as O, ipx.ops as ops, ipx.luts as luts
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
for shape in [(1, 1), (7, 1), (33, 70), (40, 342), (18, 2051), (300, 411)]:
    g = rng.integers(0, 256, shape, dtype=np.uint8)
    c = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    tg, tc = torch.from_numpy(g).to(dev), torch.from_numpy(c).to(dev)
    lut = luts.gamma(0.6)
    assert np.array_equal(ops.apply_lut(tg, lut).cpu().numpy(), lut[g])
    assert np.array_equal(ops.apply_lut(tc, lut).cpu().numpy(), lut[c])
    assert np.array_equal(ops.hist(tg).cpu().numpy().astype(np.uint32).reshape(256), O.hist_u8(g))
    assert np.array_equal(ops.equalize_hist(tg).cpu().numpy(), O.equalize_hist_u8(g))
    gray, eq = ops.bgr2gray_equalize(tc)
    assert np.array_equal(gray.cpu().numpy(), O.bgr2gray_u8(c))
    assert np.array_equal(eq.cpu().numpy(), O.equalize_hist_u8(O.bgr2gray_u8(c)))
    assert np.array_equal(ops.bgr2gray_lut(tc, lut).cpu().numpy(), lut[O.bgr2gray_u8(c)])
print("ok")
''' % (os.path.join(here, "..", "image-processing_amd"), os.path.join(here, "..", "oracle"), here)
    env = dict(os.environ, IPX_NT_BYTES="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


@pytest.mark.parametrize("shape", [(1, 1), (3, 17), (5, 44), (9, 45), (33, 70), (40, 342), (21, 1030), (18, 2051), (130, 96)])
def test_morphology_small_elements_strip_walker(dev, shape):
    """3 x 3 / 5 x 5 elements with at most two distinct row masks run on the strip walker (ipx_conv.hip): asymmetric
    elements (orientation), empty rows, rectangles, crosses, ellipses, and an element with three masks (tile kernel);
    widths around the 44-byte vector threshold and ragged 16-byte tails; erode <= src <= dilate where the centre is set."""
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    t = dev_t(img, dev)
    ses = [np.ones((3, 3), np.uint8), np.ones((5, 5), np.uint8), ops.ellipse_se(3), ops.ellipse_se(5),
           np.array([[1, 0, 0], [1, 1, 0], [1, 0, 0]], np.uint8), np.array([[0, 0, 0], [0, 1, 1], [0, 0, 0]], np.uint8),
           np.array([[0, 0, 1, 1, 0], [1, 0, 0, 0, 1], [0, 0, 0, 0, 0], [1, 0, 0, 0, 1], [0, 0, 1, 1, 0]], np.uint8),
           np.array([[1, 0, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 1, 0, 0, 0], [1, 0, 0, 0, 0]], np.uint8),
           # centre column / full row (the specialised kernel form) with empty rows; the same two masks met in the other order
           np.array([[0, 0, 1, 0, 0], [0, 0, 0, 0, 0], [1, 1, 1, 1, 1], [0, 0, 0, 0, 0], [0, 0, 1, 0, 0]], np.uint8),
           np.array([[1, 1, 1], [0, 1, 0], [1, 1, 1]], np.uint8), np.array([[0, 0, 0], [1, 1, 1], [0, 0, 0]], np.uint8)]
    for se in ses:
        np.testing.assert_array_equal(host(ops.erode(t, se)), O.erode_u8(img, se), err_msg=f"erode {se.tolist()}")
        np.testing.assert_array_equal(host(ops.dilate(t, se)), O.dilate_u8(img, se), err_msg=f"dilate {se.tolist()}")
    batch = rng.integers(0, 256, (3,) + shape + (1,), dtype=np.uint8)
    got = host(ops.dilate(dev_t(batch, dev), ops.ellipse_se(5)))
    for i in range(3):
        np.testing.assert_array_equal(got[i, :, :, 0], O.dilate_u8(batch[i, :, :, 0], ops.ellipse_se(5)))

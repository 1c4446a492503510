"""16-bit PNG uploads (no GPU): the reference converts a 16-bit COLOUR file to gray on its 16-bit samples and shifts
afterwards (server.py:516-523), Pillow would truncate first.  Host half of that path: codec.LazyPng16, the numpy
reconstruction behind its fall-back, the oracle's restatement of cvtColor on uint16 (parity unpinned, like every cv2
call), and the dispatch in ipx.app on the checker backend."""
import io

import numpy as np
import pytest
from PIL import Image

import coracle as CO
import ipx.app as app_mod
import ipx.codec as codec
import oracle as O
from oracle_ops import OracleBackend
from png16 import expected_gray, png16


def _img(rng, ch, h=37, w=53):
    return rng.integers(0, 65536, (h, w) if ch == 1 else (h, w, ch), dtype=np.uint16)


def test_oracle_gray16_known_answers():
    # equal channels map to themselves (the coefficients sum to 2^15), extremes stay, the rounding is half-up at 2^14
    v = np.array([0, 1, 255, 256, 32767, 32768, 65534, 65535], np.uint16)
    np.testing.assert_array_equal(O.bgr2gray_u16(np.repeat(v[None, :, None], 3, 2))[0], v)
    one = np.array([[[1, 0, 0]], [[0, 1, 0]], [[0, 0, 1]], [[4, 0, 0]], [[5, 0, 0]]], np.uint16)   # B, G, R
    np.testing.assert_array_equal(O.bgr2gray_u16(one)[:, 0], [0, 1, 0, 0, 1])    # 3735 / 19235 / 9798 / 14940 / 18675 vs 16384
    rng = np.random.default_rng(1)
    a = rng.integers(0, 65536, (64, 64, 3), dtype=np.uint16)
    f = 0.114 * a[..., 0] + 0.587 * a[..., 1] + 0.299 * a[..., 2]                 # README.MD:70 weights
    assert np.abs(O.bgr2gray_u16(a).astype(np.float64) - f).max() <= 0.5 + 65535 * 3e-5
    # the order matters: converting the high bytes with the 8-bit formula is NOT the same image
    hi = O.bgr2gray_u8((a >> 8).astype(np.uint8))
    lo = O.gray_u8_from_u16(a)
    d = np.abs(hi.astype(int) - lo.astype(int))
    assert d.max() == 1 and 0.05 < (d > 0).mean() < 0.5


@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_host_reconstruction_equals_c_oracle_for_every_filter(ch):
    rng = np.random.default_rng(ch)
    a = _img(rng, ch, 23, 41)
    for types in (np.arange(23) % 5, rng.integers(0, 5, 23), np.full(23, 4), np.full(23, 3)):
        blob = png16(a, types)
        w, h, c, depth, raw = codec.png_scanlines(blob)
        assert (w, h, c, depth) == (41, 23, ch, 16)
        got = codec.png_unfilter_host(raw, h, w, 2 * c)
        np.testing.assert_array_equal(got, CO.png_unfilter(np.frombuffer(raw, np.uint8), h, w, 2 * c))
        np.testing.assert_array_equal(got, np.frombuffer(a.astype(">u2").tobytes(), np.uint8).reshape(h, -1))


def test_pillow_reads_the_test_files_as_the_same_samples():
    """the writer of these tests against an independent decoder: Pillow keeps 16 bits for gray files and the high bytes for
    colour ones"""
    rng = np.random.default_rng(9)
    g = _img(rng, 1)
    np.testing.assert_array_equal(np.array(Image.open(io.BytesIO(png16(g)))), g)
    c = _img(rng, 3)
    np.testing.assert_array_equal(np.array(Image.open(io.BytesIO(png16(c)))), (c >> 8).astype(np.uint8))


@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_lazy16_through_the_dispatch_on_the_checker_backend(ch):
    rng = np.random.default_rng(10 + ch)
    a = _img(rng, ch)
    blob = png16(a)
    lz = codec.decode_lazy(blob, allow16=True)
    assert isinstance(lz, codec.LazyPng16) and (lz.h, lz.w, lz.channels, lz.bytes_per_pixel) == (37, 53, ch, 2 * ch)
    assert not isinstance(codec.decode_lazy(blob), codec.LazyPng16)       # callers that did not ask get Pillow's array
    exp = expected_gray(a)
    np.testing.assert_array_equal(app_mod.gray_from_decoded(OracleBackend(), lz), exp)
    client = app_mod.create_app(backend=OracleBackend()).test_client()

    def post(mode, **extra):
        r = client.post("/api/process", data=dict({"mode": mode, "file": (io.BytesIO(blob), "x.png")}, **extra),
                        content_type="multipart/form-data")
        assert r.status_code == 200, r.get_json()
        return codec.from_data_url_png(r.get_json()["dataURL"])
    np.testing.assert_array_equal(post("gray8"), exp)
    np.testing.assert_array_equal(post("negative"), 255 - exp)
    np.testing.assert_array_equal(post("histogram_eq"), O.equalize_hist_u8(exp))
    np.testing.assert_array_equal(post("log", c="2.0"), O.log_u8(exp, 2.0))
    np.testing.assert_array_equal(post("median", k="3"), O.median_blur_u8(exp, 3))
    np.testing.assert_array_equal(post("denoise_manual", method="median", kernel="3"), O.median_blur_u8(exp, 3))
    # IMREAD_COLOR (mode=sharpening) strips 16-bit samples to their high bytes before anything else (png_set_strip_16)
    hi = (a >> 8).astype(np.uint8)
    bgr = {1: lambda: np.repeat(hi[:, :, None], 3, 2), 2: lambda: np.repeat(hi[:, :, :1], 3, 2),
           3: lambda: hi[:, :, ::-1], 4: lambda: hi[:, :, 2::-1]}[ch]()
    np.testing.assert_array_equal(post("sharpening"), O.laplacian_sharpen_u8(np.ascontiguousarray(bgr)))


def test_batch_endpoint_decodes_16_bit_files_ahead():
    rng = np.random.default_rng(20)
    imgs = [_img(rng, 3), _img(rng, 1), _img(rng, 4)]
    client = app_mod.create_app(backend=OracleBackend()).test_client()
    for mode, f in (("gray8", lambda g: g), ("sharpening", None)):
        data = {"mode": mode, "file": [(io.BytesIO(png16(a)), f"{i}.png") for i, a in enumerate(imgs)]}
        js = client.post("/api/process_batch", data=data, content_type="multipart/form-data").get_json()
        assert js["count"] == 3 and all(r["status"] == 200 for r in js["results"])
        if f is not None:
            for a, r in zip(imgs, js["results"]):
                np.testing.assert_array_equal(codec.from_data_url_png(r["dataURL"]), expected_gray(a))

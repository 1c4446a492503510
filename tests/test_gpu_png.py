"""GPU parity of the PNG reconstruction kernel (ipx_png_unfilter_u8) through the C ABI: against the C oracle's restatement
of the PNG specification's filters (itself pinned by Pillow's decoder in tests/test_png_host.py) on scan lines filtered with
every type, and against Pillow's decode of PNG files Pillow wrote.  Byte-exact."""
import io

import numpy as np
import pytest
import torch
from PIL import Image

import coracle as CO
import ipx.codec as codec
import ipx.ops as ops

pytestmark = pytest.mark.gpu


def _unfilter(dev, raw, h, w, bpp):
    t = torch.from_numpy(np.ascontiguousarray(raw)).to(dev)
    out, status = ops.png_unfilter(t, h, w, bpp)
    torch.cuda.synchronize()
    return out.cpu().numpy(), int(status.item())


def _image(rng, h, L, kind):
    if kind == 0:
        return rng.integers(0, 256, (h, L), dtype=np.uint8)
    if kind == 1:                                                      # smooth: predictions are close, bytes wrap around 0
        yy, xx = np.mgrid[0:h, 0:L]
        return ((xx * 3 + yy * 5 + rng.integers(0, 3, (h, L))) % 256).astype(np.uint8)
    return np.where(rng.random((h, L)) < 0.5, 0, 255).astype(np.uint8)  # extremes: every Paeth tie-break


@pytest.mark.parametrize("bpp", [1, 2, 3, 4, 6, 8])
def test_every_filter_type_every_pixel_size(dev, bpp):
    """rows filtered with types 0..4 in turn, one type throughout, and at random; heights around the 64-row waves (the
    hand-over between blocks), widths around the 32-pixel publishing chunks and the 8-pixel prefetch."""
    rng = np.random.default_rng(bpp)
    for h, w in ((1, 1), (2, 2), (3, 7), (63, 31), (64, 32), (65, 33), (130, 40), (200, 100), (70, 300), (257, 41), (5, 1000)):
        for kind in range(3):
            img = _image(rng, h, w * bpp, kind)
            plans = [np.arange(h) % 5, rng.integers(0, 5, h)] + ([np.full(h, t) for t in range(5)] if h in (65, 130) else [])
            for types in plans:
                raw = CO.png_filter(img, bpp, types.astype(np.uint8))
                np.testing.assert_array_equal(CO.png_unfilter(raw, h, w, bpp), img)       # (the oracle inverts itself)
                got, status = _unfilter(dev, raw, h, w, bpp)
                assert status == 0
                np.testing.assert_array_equal(got, img, err_msg=f"{h}x{w} bpp {bpp} kind {kind} types {types[:8]}")


def test_4k_rgb_frame_random_filters(dev):
    rng = np.random.default_rng(7)
    h, w = 2160, 3840
    img = _image(rng, h, w * 3, 1)
    raw = CO.png_filter(img, 3, rng.integers(0, 5, h).astype(np.uint8))
    got, status = _unfilter(dev, raw, h, w, 3)
    assert status == 0 and (got == img).all()
    img = rng.integers(0, 256, (h, w * 3), dtype=np.uint8)
    raw = CO.png_filter(img, 3, np.full(h, 4, np.uint8))                # Paeth throughout: the longest dependency chains
    got, status = _unfilter(dev, raw, h, w, 3)
    assert status == 0 and (got == img).all()


@pytest.mark.parametrize("mode,shape", [("L", (70, 133)), ("RGB", (131, 77, 3)), ("RGBA", (65, 40, 4)), ("LA", (9, 300, 2)),
                                        ("I;16", (50, 61)), ("RGB", (1, 1, 3)), ("L", (300, 5)), ("RGB", (700, 1000, 3))])
def test_files_written_by_pillow(dev, mode, shape):
    """PNG files from Pillow's encoder (adaptive filters) -> codec.png_scanlines (host inflate) -> GPU == Pillow's decode."""
    rng = np.random.default_rng(len(mode) + shape[0])
    if mode == "I;16":
        a = (np.cumsum(rng.integers(0, 40, shape), axis=1) % 65536).astype(np.uint16)
        im = Image.fromarray(a)
    else:
        a = (np.cumsum(rng.integers(0, 9, shape), axis=1) % 256).astype(np.uint8)
        im = Image.fromarray(a, mode=mode)
    buf = io.BytesIO()
    im.save(buf, format="PNG")
    w, h, ch, depth, raw = codec.png_scanlines(buf.getvalue())
    bpp = ch * depth // 8
    got, status = _unfilter(dev, np.frombuffer(raw, np.uint8), h, w, bpp)
    assert status == 0
    ref = np.array(Image.open(io.BytesIO(buf.getvalue())))
    if depth == 16:
        got = got.reshape(h, w, 2)
        got = (got[:, :, 0].astype(np.uint16) << 8) | got[:, :, 1]
    np.testing.assert_array_equal(got.reshape(ref.shape), ref)
    np.testing.assert_array_equal(ref, a)


def test_bad_filter_byte_and_argument_errors(dev):
    import ipx._lib as L
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (10, 12), dtype=np.uint8)
    raw = CO.png_filter(img, 3, np.zeros(10, np.uint8)).copy()
    raw[3 * 13] = 9                                                    # row 3: not a filter type
    got, status = _unfilter(dev, raw, 10, 4, 3)
    assert status & 1
    t = torch.zeros(64, dtype=torch.uint8, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    for args in ((None, 5, t.data_ptr(), 4, 1, 4, 1), (t.data_ptr(), 4, t.data_ptr(), 4, 1, 4, 1),      # null, short stride
                 (t.data_ptr(), 5, t.data_ptr(), 4, 1, 4, 5), (t.data_ptr(), 5, t.data_ptr(), 4, 0, 4, 1)):  # bpp 5, no rows
        with pytest.raises(L.IpxError):
            L.call("ipx_png_unfilter_u8", None, args[0], args[1], args[2], args[3], args[4], args[5], args[6],
                   t.data_ptr(), 64, st.data_ptr())
    with pytest.raises(L.IpxError):                                    # workspace too small
        L.call("ipx_png_unfilter_u8", None, t.data_ptr(), 5, t.data_ptr(), 4, 1, 4, 1, t.data_ptr(), 8, st.data_ptr())


def test_base64_equals_the_standard_library(dev):
    """ipx_base64_encode_u8 (server.py:536): every length around the 3-byte groups and the 12-byte thread granule, all byte
    values, and a PNG-sized buffer; and GpuBackend.b64encode, which the responses go through."""
    import base64
    rng = np.random.default_rng(11)
    for n in list(range(0, 40)) + [255, 256, 257, 3071, 3072, 3073, (1 << 20) + 5, 16_000_001]:
        data = rng.integers(0, 256, n, dtype=np.uint8)
        got = ops.base64_encode(torch.from_numpy(data).to(dev)).cpu().numpy().tobytes()
        assert got == base64.b64encode(data.tobytes()), n
    every = np.arange(256, dtype=np.uint8).repeat(3)
    assert ops.base64_encode(torch.from_numpy(every).to(dev)).cpu().numpy().tobytes() == base64.b64encode(every.tobytes())
    import ipx.app as app_mod
    be = app_mod.GpuBackend(str(dev))
    for n in (10, 300_000, 5_000_003):
        blob = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert be.b64encode(blob) == base64.b64encode(blob).decode("ascii")

"""Host half of the PNG fast path (no GPU): codec.png_scanlines against Pillow, the C oracle's un-filter pinned by Pillow's
decoder on files Pillow wrote (adaptive filters: all five types occur), and the cases that must fall back to Pillow."""
import io
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

import coracle as CO
import ipx.codec as codec


def _png(a, mode=None, **kw):
    buf = io.BytesIO()
    (Image.fromarray(a, mode=mode) if mode else Image.fromarray(a)).save(buf, format="PNG", **kw)
    return buf.getvalue()


@pytest.mark.parametrize("mode,shape", [("L", (70, 133)), ("RGB", (131, 77, 3)), ("RGBA", (65, 40, 4)), ("LA", (9, 300, 2)),
                                        ("I;16", (50, 61)), ("RGB", (1, 1, 3)), ("L", (300, 5)), ("RGB", (300, 400, 3))])
def test_oracle_unfilter_equals_pillow_decode(mode, shape):
    rng = np.random.default_rng(shape[0])
    seen = set()
    for smooth in (True, False):
        if mode == "I;16":
            a = (np.cumsum(rng.integers(0, 40, shape), axis=1) % 65536 if smooth else rng.integers(0, 65536, shape)).astype(np.uint16)
        else:
            a = (np.cumsum(rng.integers(0, 9, shape), axis=1) % 256 if smooth else rng.integers(0, 256, shape)).astype(np.uint8)
        blob = _png(a, None if mode == "I;16" else mode)
        w, h, ch, depth, raw = codec.png_scanlines(blob)
        bpp = ch * depth // 8
        seen |= set(np.frombuffer(raw, np.uint8)[::1 + w * bpp].tolist())
        out = CO.png_unfilter(np.frombuffer(raw, np.uint8), h, w, bpp)
        if depth == 16:
            out = out.reshape(h, w, 2)
            out = (out[:, :, 0].astype(np.uint16) << 8) | out[:, :, 1]
        ref = np.array(Image.open(io.BytesIO(blob)))
        np.testing.assert_array_equal(out.reshape(ref.shape), ref)
        # the forward filters invert: filter(unfilter(raw)) == raw
        flat = CO.png_unfilter(np.frombuffer(raw, np.uint8), h, w, bpp)
        types = np.frombuffer(raw, np.uint8)[::1 + w * bpp]
        np.testing.assert_array_equal(CO.png_filter(flat, bpp, types), np.frombuffer(raw, np.uint8))
    if shape[0] >= 50:
        assert len(seen) >= 2, seen


def test_own_encoder_and_lazy_object():
    rng = np.random.default_rng(5)
    for a in (rng.integers(0, 256, (33, 47), dtype=np.uint8), rng.integers(0, 256, (20, 31, 3), dtype=np.uint8)):
        blob = codec.encode_png(a)
        lz = codec.decode_lazy(blob)
        assert isinstance(lz, codec.LazyPng) and lz.shape == a.shape and lz.ndim == a.ndim and lz.dtype == np.uint8
        np.testing.assert_array_equal(np.asarray(lz), a)                  # the host materialisation (Pillow)
        np.testing.assert_array_equal(lz[:5], a[:5])
        out = CO.png_unfilter(np.frombuffer(lz.raw, np.uint8), lz.h, lz.w, lz.channels)
        np.testing.assert_array_equal(out.reshape(a.shape), a)


def test_files_the_fast_path_declines():
    rng = np.random.default_rng(6)
    a = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    pal = Image.fromarray(a).convert("P")
    buf = io.BytesIO(); pal.save(buf, format="PNG")
    assert codec.png_scanlines(buf.getvalue()) is None                     # palette
    buf = io.BytesIO(); Image.fromarray(a).save(buf, format="JPEG")
    assert codec.png_scanlines(buf.getvalue()) is None                     # not a PNG
    assert isinstance(codec.decode_lazy(buf.getvalue()), np.ndarray)
    bits = Image.fromarray((a[:, :, 0] > 128)).convert("1")
    buf = io.BytesIO(); bits.save(buf, format="PNG")
    assert codec.png_scanlines(buf.getvalue()) is None                     # 1 bit per pixel
    blob = _png(a)
    assert codec.png_scanlines(blob[:len(blob) // 2]) is None              # truncated
    assert codec.png_scanlines(blob[:-12]) is None                         # no IEND
    # 16-bit gray keeps Pillow's path in decode_lazy (uint16 arithmetic happens on the host, server.py:521-523)
    g16 = _png(rng.integers(0, 65536, (12, 13)).astype(np.uint16))
    got = codec.decode_lazy(g16)
    assert isinstance(got, np.ndarray) and got.dtype == np.uint16
    # a filter type byte above 4: rebuilt stream with row 2 marked 7
    w, h, ch, depth, raw = codec.png_scanlines(blob)
    bad = bytearray(raw); bad[2 * (1 + w * ch)] = 7
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(data, zlib.crc32(tag)) & 0xffffffff)
    rebuilt = blob[:33] + chunk(b"IDAT", zlib.compress(bytes(bad))) + chunk(b"IEND", b"")
    assert codec.png_scanlines(rebuilt) is None
    with pytest.raises(RuntimeError, match="Cannot decode image"):
        codec.decode_lazy(rebuilt)                                          # Pillow refuses it too: the reference's 500 text
    # interlaced
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format="PNG")
    il = bytearray(buf.getvalue()); il[28] = 1
    assert codec.png_scanlines(bytes(il)) is None

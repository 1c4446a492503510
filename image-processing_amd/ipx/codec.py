"""Host-side staging: bytes -> numpy (decode) and numpy -> PNG data URL (encode).

Mirrors server.py:492-538 with Pillow instead of cv2.imdecode/imencode (decode stays on the CPU by design,
SURVEY.md section 2 C2/C3; only the pixel arithmetic moves to the GPU).  Channel order handed to the GPU
kernels is BGR, as OpenCV decodes, so that `bgr2gray` applies OpenCV's coefficients to the right channels.
Encoded PNG bytes need not match the reference's; decoded pixels must.
"""
from __future__ import annotations

import base64
import concurrent.futures
import io
import os
import struct
import threading
import zlib

import numpy as np
from PIL import Image, ImageOps


def _open(img_bytes: bytes, exif_orientation: bool = False) -> Image.Image:
    try:
        pre = getattr(_TLS, "prefetched", None)
        if pre is not None and pre.blob is img_bytes:
            im = pre.fut.result()                                     # decoded ahead by the pool (raises what it raised)
            if isinstance(im, LazyPng16):                             # (a caller that wants what Pillow makes of the file)
                im = Image.open(io.BytesIO(im.blob))
                im.load()
            elif isinstance(im, LazyPng):                             # (a caller that wants the pixels on the host)
                im = Image.fromarray(im.array())
        else:
            im = Image.open(io.BytesIO(img_bytes))
            im.load()
        if exif_orientation:
            im = ImageOps.exif_transpose(im)
        return im
    except Exception:
        # cv2.imdecode returns None (it does not raise) for bytes it cannot decode, so server.py:531 always ends in
        # this fixed text; Pillow's own message would carry a memory address and differ from request to request
        raise RuntimeError("Cannot decode image to grayscale 8-bit. unknown error")


class DecodeAhead:
    """Decodes a list of uploads on the codec pool ahead of their consumers (/api/process_batch).  `with d.use(i):`
    makes the calling thread's next _open() of blob i return the image decoded by the pool (or re-raise its error)
    instead of decoding again; everything else about the dispatch stays as it is."""

    def __init__(self, blobs):
        self.blobs = blobs
        self.futures = [_pool().submit(self._decode, b) for b in blobs]

    @staticmethod
    def _decode(b):
        got = png_scanlines(b)                                        # inflate only; the GPU un-filters (decode_lazy)
        if got is not None and got[3] == 8 and got[2] in (1, 3, 4):
            return LazyPng(b, got[0], got[1], got[2], got[4])
        if got is not None and got[3] == 16:
            return LazyPng16(b, got[0], got[1], got[2], got[4])
        im = Image.open(io.BytesIO(b))
        im.load()
        return im

    def use(self, i):
        return _Prefetched(self.blobs[i], self.futures[i])


class _Prefetched:
    def __init__(self, blob, fut):
        self.blob, self.fut = blob, fut

    def __enter__(self):
        _TLS.prefetched = self
        return self

    def __exit__(self, *a):
        _TLS.prefetched = None
        return False


_TLS = threading.local()


# ---- PNG fast path: inflate here, un-filter on the GPU ---------------------------------------------------------------
_PNG_SIG = b"\x89PNG\r\n\x1a\n"
_PNG_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}                 # colour type -> samples per pixel (palette images: Pillow)


def png_scanlines(img_bytes: bytes):
    """Parses a non-interlaced 8- or 16-bit gray / gray+alpha / RGB / RGBA PNG and inflates its IDAT stream (zlib releases the
    GIL): returns (w, h, channels, bit depth, filtered scan lines as bytes), or None for anything else -- palette, sub-byte
    depths, Adam7, colour-key transparency (tRNS), an eXIf orientation chunk, a damaged stream, a filter type byte above 4:
    the caller then takes Pillow's decoder, which either handles the case or raises what it raises.  Chunk CRCs are not
    checked here (a stream that inflates to exactly h scan lines with valid filter types is taken as it is)."""
    b = img_bytes
    if len(b) < 33 or b[:8] != _PNG_SIG or b[12:16] != b"IHDR":
        return None
    w, h, depth, color, comp, filt, lace = struct.unpack(">IIBBBBB", b[16:29])
    if depth not in (8, 16) or color not in _PNG_CHANNELS or comp or filt or lace or w == 0 or h == 0:
        return None
    pos, idat, n = 33, [], len(b)
    while pos + 8 <= n:
        ln = int.from_bytes(b[pos:pos + 4], "big")
        tag = b[pos + 4:pos + 8]
        if tag == b"IDAT":
            idat.append(b[pos + 8:pos + 8 + ln])
        elif tag in (b"tRNS", b"eXIf", b"acTL"):
            return None
        elif tag == b"IEND":
            break
        pos += 12 + ln
    else:
        return None                                                   # no IEND: truncated
    ch = _PNG_CHANNELS[color]
    stride = 1 + w * ch * (depth // 8)
    try:
        d = zlib.decompressobj()
        raw = d.decompress(idat[0] if len(idat) == 1 else b"".join(idat), h * stride + 1)
        if len(raw) != h * stride or not d.eof and d.decompress(d.unconsumed_tail, 1):
            return None
    except zlib.error:
        return None
    if np.frombuffer(raw, np.uint8)[::stride].max() > 4:
        return None
    return w, h, ch, depth, raw


class LazyPng:
    """An 8-bit gray / RGB / RGBA PNG decoded as far as the host goes: inflated, not yet un-filtered.  Looks like the array
    decode_unchanged(..., "rgb") would return (shape, ndim, dtype); GpuBackend uploads `raw` and reconstructs the pixels
    on the GPU (ipx_png_unfilter_u8).  Anything that needs the pixels on the host (indexing, numpy functions) gets them from
    Pillow's decoder, once."""

    dtype = np.dtype(np.uint8)

    def __init__(self, blob: bytes, w: int, h: int, channels: int, raw: bytes):
        self.blob, self.w, self.h, self.channels, self.raw = blob, w, h, channels, raw
        self.bytes_per_pixel = channels
        self.shape = (h, w) if channels == 1 else (h, w, channels)
        self.ndim = len(self.shape)
        self.size = h * w * channels
        self._array = None

    def array(self) -> np.ndarray:
        if self._array is None:
            im = Image.open(io.BytesIO(self.blob))
            im.load()
            self._array = np.array(im, dtype=np.uint8)
            if self._array.shape != self.shape:
                raise RuntimeError("Cannot decode image to grayscale 8-bit. unknown error")
        return self._array

    def __array__(self, dtype=None, copy=None):
        a = self.array()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, idx):
        return self.array()[idx]


def decode_lazy(img_bytes: bytes, exif_orientation: bool = False, allow16: bool = False):
    """decode_unchanged(img_bytes, "rgb") for callers that hand the result to GpuBackend.run: an 8-bit gray / RGB / RGBA PNG
    comes back as a LazyPng (un-filtered on the GPU), everything else as the numpy array Pillow decodes.  allow16: a 16-bit
    PNG comes back as a LazyPng16 (the caller turns it into gray with ops.gray_from_u16: app.gray_from_decoded)."""
    pre = getattr(_TLS, "prefetched", None)
    got = None
    if pre is not None and pre.blob is img_bytes:
        res = pre.fut.result() if pre.fut.exception() is None else None    # (an error is re-raised, in the reference's
        if isinstance(res, LazyPng16):                                     # words, by _open below)
            if allow16:
                return res
        elif isinstance(res, LazyPng):
            return res
    else:
        got = png_scanlines(img_bytes)
    if got is not None and got[3] == 8 and got[2] in (1, 3, 4):
        w, h, ch, _, raw = got
        return LazyPng(img_bytes, w, h, ch, raw)
    if got is not None and got[3] == 16 and allow16:
        return LazyPng16(img_bytes, got[0], got[1], got[2], got[4])
    return decode_unchanged(img_bytes, "rgb", exif_orientation)


def png_unfilter_host(raw: bytes, h: int, w: int, bpp: int) -> np.ndarray:
    """PNG reconstruction (specification section 9) on the host, numpy: (h, w * bpp) uint8.  Only what LazyPng16.array() needs --
    the fall-back when the GPU reconstruction reports a problem (never seen) and test backends without a GPU; rows filtered
    with Average or Paeth cost a Python loop over their pixels."""
    L = w * bpp
    a = np.frombuffer(raw, np.uint8, count=h * (L + 1)).reshape(h, L + 1)
    out = np.zeros((h, L), np.uint8)
    zero = np.zeros(L, np.uint8)
    for y in range(h):
        t, f = int(a[y, 0]), a[y, 1:]
        up = out[y - 1] if y else zero
        if t == 0:
            out[y] = f
        elif t == 1:                                                   # Sub: a running sum per byte lane, modulo 256
            out[y] = np.cumsum(f.reshape(w, bpp), axis=0, dtype=np.uint8).reshape(L)
        elif t == 2:
            out[y] = f + up
        elif t in (3, 4):
            fr, ur, o = f.reshape(w, bpp).astype(np.int32), up.reshape(w, bpp).astype(np.int32), np.zeros((w, bpp), np.int32)
            left, ul = np.zeros(bpp, np.int32), np.zeros(bpp, np.int32)
            for x in range(w):
                if t == 3:
                    pred = (left + ur[x]) >> 1
                else:
                    p = left + ur[x] - ul
                    pa, pb, pc = np.abs(p - left), np.abs(p - ur[x]), np.abs(p - ul)
                    pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, ur[x], ul))
                left = o[x] = (fr[x] + pred) & 255
                ul = ur[x]
            out[y] = o.reshape(L).astype(np.uint8)
        else:
            raise RuntimeError("Cannot decode image to grayscale 8-bit. unknown error")
    return out


class LazyPng16:
    """A 16-bit gray / gray+alpha / RGB / RGBA PNG, inflated but not yet un-filtered (decode_lazy(..., allow16=True)).  The
    reference gray-converts such a file on its 16-bit samples and shifts afterwards (server.py:516-523); Pillow would hand over
    8-bit RGB.  GpuBackend uploads `raw`, reconstructs the scan lines on the GPU and gives the consumer an (h, 2 * w * channels)
    uint8 tensor of big-endian samples in file order -- the input of ops.gray_from_u16.  array() is the same matrix made on the
    host."""

    depth = 16

    def __init__(self, blob: bytes, w: int, h: int, channels: int, raw: bytes):
        self.blob, self.w, self.h, self.channels, self.raw = blob, w, h, channels, raw
        self.bytes_per_pixel = 2 * channels
        self.shape = (h, w * self.bytes_per_pixel)
        self._array = None

    def array(self) -> np.ndarray:
        if self._array is None:
            self._array = png_unfilter_host(self.raw, self.h, self.w, self.bytes_per_pixel)
        return self._array

    def __array__(self, dtype=None, copy=None):
        a = self.array()
        return a if dtype is None else a.astype(dtype, copy=False)


def decode_unchanged(img_bytes: bytes, order: str = "bgr", exif_orientation: bool = False) -> np.ndarray:
    """cv2.imdecode(arr, IMREAD_UNCHANGED) equivalent (server.py:499-501): 2-D gray (uint8/uint16) or
    H x W x {3,4} BGR(A) uint8.  order="rgb" skips the channel reversal (a 25 MB host copy at 4K): the caller then
    reverses on the GPU, or not at all for per-channel filters.  IMREAD_UNCHANGED ignores the EXIF orientation tag
    (exif_orientation=False); IMREAD_COLOR applies it (decode_color).
    A 16-bit COLOUR file reaches the reference as uint16 BGR (gray conversion on 16 bits, then >> 8, server.py:517-523);
    Pillow hands over 8-bit RGB.  For PNG files the gray paths therefore go through decode_lazy(..., allow16=True) and
    ops.gray_from_u16 instead of this function; what remains here is a 16-bit colour file in another container (TIFF),
    where the 8-bit formula runs on truncated samples and a pixel can differ by one."""
    im = _open(img_bytes, exif_orientation)
    if im.mode in ("L",):
        return np.array(im, dtype=np.uint8)
    if im.mode in ("I;16", "I;16B", "I;16L", "I"):
        return np.array(im).astype(np.uint16)
    if im.mode == "1":
        return (np.array(im, dtype=np.uint8) * 255).astype(np.uint8)
    if im.mode in ("RGBA", "LA", "PA") or (im.mode == "P" and "transparency" in im.info):
        a = np.array(im.convert("RGBA"), dtype=np.uint8)
        return a if order == "rgb" else np.ascontiguousarray(a[:, :, [2, 1, 0, 3]])   # RGBA -> BGRA
    a = np.array(im.convert("RGB"), dtype=np.uint8)
    return a if order == "rgb" else np.ascontiguousarray(a[:, :, ::-1])               # RGB -> BGR


def decode_color(img_bytes: bytes, order: str = "bgr"):
    """cv2.imdecode(arr, IMREAD_COLOR) equivalent (server.py:252): always H x W x 3 BGR uint8, or None.  Unlike
    IMREAD_UNCHANGED this flag honours the EXIF orientation of a JPEG (a phone photo comes out upright)."""
    try:
        a = decode_unchanged(img_bytes, order, exif_orientation=True)
    except RuntimeError:
        return None
    if a.dtype == np.uint16:
        a = (a >> 8).astype(np.uint8)
    if a.ndim == 2:
        return np.ascontiguousarray(np.repeat(a[:, :, None], 3, axis=2))
    return a if a.shape[2] == 3 else np.ascontiguousarray(a[:, :, :3])


def decode_color_lazy(img_bytes: bytes):
    """decode_color(img_bytes, "rgb") for GpuBackend.run: an 8-bit RGB PNG without an orientation chunk comes back as a
    LazyPng, everything else as decode_color returns it (H x W x 3 uint8, or None)."""
    pre = getattr(_TLS, "prefetched", None)
    if pre is not None and pre.blob is img_bytes:
        res = pre.fut.result() if pre.fut.exception() is None else None
        if isinstance(res, LazyPng) and res.channels == 3:
            return res
    else:
        got = png_scanlines(img_bytes)
        if got is not None and got[3] == 8 and got[2] == 3:
            return LazyPng(img_bytes, got[0], got[1], 3, got[4])
    return decode_color(img_bytes, "rgb")


_POOL = None
_POOL_LOCK = threading.Lock()
_BAND_BYTES = 1 << 20            # filtered bytes per independently deflated band


def _pool():
    global _POOL
    with _POOL_LOCK:
        if _POOL is None:
            n = int(os.environ.get("IPX_CODEC_THREADS", "0")) or min(16, os.cpu_count() or 1)
            _POOL = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, n), thread_name_prefix="ipx-png")
        return _POOL


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(data, zlib.crc32(tag)) & 0xFFFFFFFF)


def adler32_combine(a1: int, a2: int, len2: int) -> int:
    """Adler-32 of two concatenated blocks from the checksums of each and the length of the second (zlib's
    adler32_combine, which Python's zlib module does not expose): lets the bands be summed on the pool."""
    base = 65521
    rem = len2 % base
    s1 = a1 & 0xFFFF
    s2 = (rem * s1) % base
    s1 += (a2 & 0xFFFF) + base - 1
    s2 += ((a1 >> 16) & 0xFFFF) + ((a2 >> 16) & 0xFFFF) + base - rem
    if s1 >= base:
        s1 -= base
    if s1 >= base:
        s1 -= base
    if s2 >= (base << 1):
        s2 -= base << 1
    if s2 >= base:
        s2 -= base
    return s1 | (s2 << 16)


def encode_png(img: np.ndarray, level: int = 1) -> bytes:
    """PNG writer for H x W (gray) or H x W x 3 (RGB order) uint8 whose deflate work runs on all host cores: the
    'Up'-filtered scan lines are cut into bands, every band is deflated on its own (zlib releases the GIL) and ends on a
    byte boundary (Z_SYNC_FLUSH), and the bands are concatenated into ONE zlib stream, as pigz does.  Each band becomes an
    IDAT chunk of its own, so that its CRC-32 and its share of the stream's Adler-32 are computed by the same pool thread
    (28 ms of serial checksumming per 4K colour frame otherwise); the stream's 2-byte header and 4-byte trailer are two
    tiny IDAT chunks around them.  PNG encoding is the largest part of a request (scripts/e2ebench.py); the decoded
    pixels are what the contract fixes, not these bytes."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    rows = img.reshape(h, -1)
    raw = np.empty((h, rows.shape[1] + 1), np.uint8)
    raw[:, 0] = 2                                                      # filter type 2 (Up) on every row
    raw[0, 1:] = rows[0]
    np.subtract(rows[1:], rows[:-1], out=raw[1:, 1:])                  # uint8 arithmetic wraps, as the filter does
    flat = raw.reshape(-1)
    per = max(1, _BAND_BYTES // raw.shape[1]) * raw.shape[1]
    bands = [flat[o:o + per] for o in range(0, flat.size, per)]

    def deflate(i):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        data = co.compress(bands[i]) + co.flush(zlib.Z_FINISH if i == len(bands) - 1 else zlib.Z_SYNC_FLUSH)
        return _chunk(b"IDAT", data), zlib.adler32(bands[i]), bands[i].size

    parts = [deflate(0)] if len(bands) == 1 else list(_pool().map(deflate, range(len(bands))))
    adler = 1
    for _, a, n in parts:
        adler = adler32_combine(adler, a, n)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 0 if img.ndim == 2 else 2, 0, 0, 0)
    return b"".join([_PNG_SIG, _chunk(b"IHDR", ihdr), _chunk(b"IDAT", b"\x78\x01")] + [p[0] for p in parts] +
                    [_chunk(b"IDAT", struct.pack(">I", adler & 0xFFFFFFFF)), _chunk(b"IEND", b"")])


def to_data_url_png(img: np.ndarray, order: str = "bgr", b64=None) -> str:
    """server.py:533-538.  `img` is H x W gray or H x W x 3 uint8, BGR unless order="rgb".  b64: bytes -> ASCII str
    (GpuBackend.b64encode); default: the host's base64 module."""
    if b64 is None:
        b64 = lambda data: base64.b64encode(data).decode("ascii")
    img = np.asarray(img)
    if img.ndim == 3 and img.shape[2] == 1:
        img = img[:, :, 0]
    if img.ndim == 3 and order != "rgb":
        img = img[:, :, ::-1]                                          # BGR -> RGB, PNG's order
    if img.dtype != np.uint8 or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] != 3) or img.size == 0:
        buf = io.BytesIO()                                             # anything unusual: Pillow's writer
        Image.fromarray(np.ascontiguousarray(img)).save(buf, format="PNG", compress_level=1)
        return "data:image/png;base64," + b64(buf.getvalue())
    return "data:image/png;base64," + b64(encode_png(img))


def from_data_url_png(url: str) -> np.ndarray:
    """Inverse of to_data_url_png (tests, clients): returns gray H x W or BGR H x W x 3."""
    assert url.startswith("data:image/png;base64,")
    im = Image.open(io.BytesIO(base64.b64decode(url.split(",", 1)[1])))
    a = np.array(im)
    return a if a.ndim == 2 else np.ascontiguousarray(a[:, :, ::-1])

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


def decode_unchanged(img_bytes: bytes, order: str = "bgr", exif_orientation: bool = False) -> np.ndarray:
    """cv2.imdecode(arr, IMREAD_UNCHANGED) equivalent (server.py:499-501): 2-D gray (uint8/uint16) or
    H x W x {3,4} BGR(A) uint8.  order="rgb" skips the channel reversal (a 25 MB host copy at 4K): the caller then
    reverses on the GPU, or not at all for per-channel filters.  IMREAD_UNCHANGED ignores the EXIF orientation tag
    (exif_orientation=False); IMREAD_COLOR applies it (decode_color).
    Known divergence: a 16-bit COLOUR file reaches the reference as uint16 BGR (gray conversion on 16 bits, then
    >> 8, server.py:517-523); Pillow hands over 8-bit RGB, so the 8-bit gray formula runs on truncated samples and a
    pixel can differ by one."""
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


def encode_png(img: np.ndarray, level: int = 1) -> bytes:
    """PNG writer for H x W (gray) or H x W x 3 (RGB order) uint8 whose deflate work runs on all host cores: the
    'Up'-filtered scan lines are cut into bands, every band is deflated on its own (zlib releases the GIL) and ends on a
    byte boundary (Z_SYNC_FLUSH), and the bands are concatenated into ONE zlib stream, as pigz does.  PNG encoding is
    the largest part of a request (scripts/e2ebench.py); the decoded pixels are what the contract fixes, not these bytes."""
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
        return co.compress(bands[i]) + co.flush(zlib.Z_FINISH if i == len(bands) - 1 else zlib.Z_SYNC_FLUSH)

    parts = [deflate(0)] if len(bands) == 1 else list(_pool().map(deflate, range(len(bands))))
    stream = b"\x78\x01" + b"".join(parts) + struct.pack(">I", zlib.adler32(flat) & 0xFFFFFFFF)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 0 if img.ndim == 2 else 2, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", stream) + _chunk(b"IEND", b"")


def to_data_url_png(img: np.ndarray, order: str = "bgr") -> str:
    """server.py:533-538.  `img` is H x W gray or H x W x 3 uint8, BGR unless order="rgb"."""
    img = np.asarray(img)
    if img.ndim == 3 and img.shape[2] == 1:
        img = img[:, :, 0]
    if img.ndim == 3 and order != "rgb":
        img = img[:, :, ::-1]                                          # BGR -> RGB, PNG's order
    if img.dtype != np.uint8 or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] != 3) or img.size == 0:
        buf = io.BytesIO()                                             # anything unusual: Pillow's writer
        Image.fromarray(np.ascontiguousarray(img)).save(buf, format="PNG", compress_level=1)
        return "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode("ascii")
    return "data:image/png;base64," + base64.b64encode(encode_png(img)).decode("ascii")


def from_data_url_png(url: str) -> np.ndarray:
    """Inverse of to_data_url_png (tests, clients): returns gray H x W or BGR H x W x 3."""
    assert url.startswith("data:image/png;base64,")
    im = Image.open(io.BytesIO(base64.b64decode(url.split(",", 1)[1])))
    a = np.array(im)
    return a if a.ndim == 2 else np.ascontiguousarray(a[:, :, ::-1])

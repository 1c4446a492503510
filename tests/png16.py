"""TEST helper: a 16-bit PNG writer (Pillow cannot write 16-bit RGB / RGBA files).  Rows are filtered with the C oracle's
forward filters (types 0..4 in turn unless given), deflated in one IDAT chunk."""
import struct
import zlib

import numpy as np

import coracle as CO


def png16(a: np.ndarray, types=None) -> bytes:
    """a: uint16, H x W (gray) or H x W x {2, 3, 4} (gray+alpha, RGB, RGBA in FILE order)."""
    assert a.dtype == np.uint16
    h, w = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    color = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    rows = np.frombuffer(a.astype(">u2").tobytes(), np.uint8).reshape(h, w * ch * 2)
    if types is None:
        types = np.arange(h) % 5
    raw = CO.png_filter(rows, ch * 2, np.asarray(types, np.uint8))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, color, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw.tobytes(), 1)) + chunk(b"IEND", b""))


def expected_gray(a: np.ndarray) -> np.ndarray:
    """What the reference's ensure_gray_uint8_from_bytes makes of that file (server.py:509-523), by the oracle: cv2 decodes to
    B, G, R(, A) uint16, converts on 16 bits, shifts."""
    import oracle as O
    if a.ndim == 2:
        return O.u16_to_u8_shr8(a)
    if a.shape[2] == 2:                                               # gray + alpha: B = G = R = gray
        return O.u16_to_u8_shr8(np.ascontiguousarray(a[..., 0]))
    return O.gray_u8_from_u16(np.ascontiguousarray(a[..., [2, 1, 0]]))

"""Host-side staging: bytes -> numpy (decode) and numpy -> PNG data URL (encode).

Mirrors server.py:492-538 with Pillow instead of cv2.imdecode/imencode (decode stays on the CPU by design,
SURVEY.md section 2 C2/C3; only the pixel arithmetic moves to the GPU).  Channel order handed to the GPU
kernels is BGR, as OpenCV decodes, so that `bgr2gray` applies OpenCV's coefficients to the right channels.
Encoded PNG bytes need not match the reference's; decoded pixels must.
"""
from __future__ import annotations

import base64
import io

import numpy as np
from PIL import Image


def _open(img_bytes: bytes) -> Image.Image:
    try:
        im = Image.open(io.BytesIO(img_bytes))
        im.load()
        return im
    except Exception as e:  # same message family as server.py:531
        raise RuntimeError("Cannot decode image to grayscale 8-bit. " + str(e))


def decode_unchanged(img_bytes: bytes) -> np.ndarray:
    """cv2.imdecode(arr, IMREAD_UNCHANGED) equivalent (server.py:499-501): 2-D gray (uint8/uint16) or
    H x W x {3,4} BGR(A) uint8."""
    im = _open(img_bytes)
    if im.mode in ("L",):
        return np.array(im, dtype=np.uint8)
    if im.mode in ("I;16", "I;16B", "I;16L", "I"):
        return np.array(im).astype(np.uint16)
    if im.mode == "1":
        return (np.array(im, dtype=np.uint8) * 255).astype(np.uint8)
    if im.mode in ("RGBA", "LA", "PA") or (im.mode == "P" and "transparency" in im.info):
        a = np.array(im.convert("RGBA"), dtype=np.uint8)
        return np.ascontiguousarray(a[:, :, [2, 1, 0, 3]])            # RGBA -> BGRA
    a = np.array(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(a[:, :, ::-1])                        # RGB -> BGR


def decode_color(img_bytes: bytes):
    """cv2.imdecode(arr, IMREAD_COLOR) equivalent (server.py:252): always H x W x 3 BGR uint8, or None."""
    try:
        a = decode_unchanged(img_bytes)
    except RuntimeError:
        return None
    if a.dtype == np.uint16:
        a = (a >> 8).astype(np.uint8)
    if a.ndim == 2:
        return np.ascontiguousarray(np.repeat(a[:, :, None], 3, axis=2))
    return np.ascontiguousarray(a[:, :, :3])


def to_data_url_png(img: np.ndarray) -> str:
    """server.py:533-538.  `img` is H x W gray or H x W x 3 BGR uint8."""
    img = np.asarray(img)
    if img.ndim == 3:
        img = img[:, :, ::-1]                                          # BGR -> RGB for Pillow
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(img)).save(buf, format="PNG", compress_level=1)
    return "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode("ascii")


def from_data_url_png(url: str) -> np.ndarray:
    """Inverse of to_data_url_png (tests, clients): returns gray H x W or BGR H x W x 3."""
    assert url.startswith("data:image/png;base64,")
    im = Image.open(io.BytesIO(base64.b64decode(url.split(",", 1)[1])))
    a = np.array(im)
    return a if a.ndim == 2 else np.ascontiguousarray(a[:, :, ::-1])

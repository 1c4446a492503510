"""ctypes view of oracle/_build/liboracle.so (the C restatement) -- TEST INFRASTRUCTURE ONLY.

Used by tests (cross-check of the two restatements) and by bench.py's cpu_baseline leg
(kind "port").  Build with `make -C oracle` (done by `__graft_entry__.build()`).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError(f"{_PATH} missing: run `make -C oracle`")
        _lib = C.CDLL(_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _cn(img):
    return 1 if img.ndim == 2 else img.shape[2]


def bgr2gray_u8(bgr):
    bgr = np.ascontiguousarray(bgr)
    h, w, cn = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray_u8(_p(bgr), _p(out), h, w, cn)
    return out


def lut_u8(img, lut):
    img = np.ascontiguousarray(img)
    lut = np.ascontiguousarray(lut, np.uint8)
    out = np.empty_like(img)
    lib().orc_lut_u8(_p(img), _p(out), C.c_size_t(img.size), _p(lut))
    return out


def hist_u8(img):
    img = np.ascontiguousarray(img)
    h = np.zeros(256, np.uint32)
    lib().orc_hist_u8(_p(img), C.c_size_t(img.size), _p(h))
    return h


def equalize_lut_from_hist(hist):
    hist = np.ascontiguousarray(hist, np.uint32)
    lut = np.zeros(256, np.uint8)
    lib().orc_equalize_lut(_p(hist), _p(lut))
    return lut


def equalize_hist_u8(img):
    img = np.ascontiguousarray(img)
    out = np.empty_like(img)
    lib().orc_equalize_hist_u8(_p(img), _p(out), C.c_size_t(img.size))
    return out


def _conv3(img, op):
    img = np.ascontiguousarray(img)
    out = np.empty_like(img)
    lib().orc_conv3_u8(_p(img), _p(out), img.shape[0], img.shape[1], _cn(img), op)
    return out


def laplacian_sharpen_u8(img):
    return _conv3(img, 0)


def box3_u8(img):
    return _conv3(img, 1)


def sobel_u8(img):
    return _conv3(img, 2)


def gauss_kernel_q8(k, sigma=0.0):
    out = np.zeros(k, np.uint16)
    rc = lib().orc_gauss_kernel_q8(int(k), C.c_double(sigma), _p(out))
    if rc != 0:
        raise ValueError("bad kernel size")
    return out


def sep_filter_q8_u8(img, kx, ky):
    img = np.ascontiguousarray(img)
    kx = np.ascontiguousarray(kx, np.uint16)
    ky = np.ascontiguousarray(ky, np.uint16)
    out = np.empty_like(img)
    lib().orc_sep_q8_u8(_p(img), _p(out), img.shape[0], img.shape[1], _cn(img),
                        _p(kx), len(kx), _p(ky), len(ky))
    return out


def gaussian_blur_u8(img, k, sigma=0.0):
    q = gauss_kernel_q8(k, sigma)
    return sep_filter_q8_u8(img, q, q)


def png_unfilter(raw, h, w, bpp):
    """PNG reconstruction of h inflated scan lines ([type][w*bpp bytes] each) -> (h, w*bpp) uint8; raises on a bad type byte."""
    raw = np.ascontiguousarray(raw, np.uint8).reshape(-1)
    stride = 1 + w * bpp
    assert raw.size >= h * stride
    out = np.empty((h, w * bpp), np.uint8)
    rc = lib().orc_png_unfilter(_p(raw), C.c_size_t(stride), _p(out), int(h), int(w), int(bpp))
    if rc:
        raise ValueError(f"row {rc - 1}: filter type above 4")
    return out


def png_filter(img, bpp, types):
    """Forward PNG filters: img (h, w*bpp) uint8, types[y] in 0..4 -> the filtered scan lines, flat (h * (1 + w*bpp))."""
    img = np.ascontiguousarray(img, np.uint8)
    h, L = img.shape
    types = np.ascontiguousarray(types, np.uint8)
    assert types.size == h and L % bpp == 0
    raw = np.empty((h, 1 + L), np.uint8)
    lib().orc_png_filter(_p(img), _p(raw), C.c_size_t(1 + L), int(h), int(L // bpp), int(bpp), _p(types))
    return raw.reshape(-1)

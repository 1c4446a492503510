"""The C-ABI library loads and exports every symbol include/ipx.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ipx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ipx_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ("ipx_bgr2gray_u8", "ipx_lut_u8", "ipx_hist_u8", "ipx_equalize_hist_u8", "ipx_laplacian_sharpen_u8",
              "ipx_box3_u8", "ipx_sobel_u8", "ipx_sep_filter_q8_u8", "ipx_gaussian_blur_u8", "ipx_last_error"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    import ipx._lib as L
    lib = L.load()
    for s in declared_symbols():
        assert hasattr(lib, s), f"libipx.so does not export {s}"
    assert set(L.SIGNATURES) == set(declared_symbols()), "binding table and header disagree"
    assert lib.ipx_version() >= 100


def test_gauss_kernel_helper_matches_oracle():
    import ipx.ops as ops
    import oracle
    for k in (1, 3, 5, 7, 9, 11, 13, 15, 21, 31, 63, 127, 255):
        np.testing.assert_array_equal(ops.gauss_kernel_q8(k), oracle.gauss_kernel_q8(k))
    for k, s in ((5, 1.0), (9, 2.5), (15, 0.7), (3, 10.0)):
        np.testing.assert_array_equal(ops.gauss_kernel_q8(k, s), oracle.gauss_kernel_q8(k, s))


def test_argument_errors_are_reported_not_thrown():
    import ipx._lib as L
    lib = L.load()
    buf = (ctypes.c_uint16 * 4)()
    assert lib.ipx_gauss_kernel_q8(4, 0.0, ctypes.addressof(buf)) == L.IPX_ERR_INVALID
    assert b"odd" in lib.ipx_last_error()
    # null pointers / bad sizes are rejected before any HIP call
    assert lib.ipx_lut_u8(None, None, None, 1, 4, 4, 1, 4, 16, 4, 16, None) == L.IPX_ERR_INVALID
    assert lib.ipx_box3_u8(None, 1, 1, 1, 0, 4, 1, 4, 16, 4, 16, None, None) == L.IPX_ERR_INVALID
    assert lib.ipx_sobel_u8(None, 1, 1, 1, 4, 4, 5, 20, 80, 20, 80, None, None) == L.IPX_ERR_UNSUPPORTED
    assert lib.ipx_bgr2gray_u8(None, 1, 1, 1, 4, 4, 3, 8, 48, 4, 16) == L.IPX_ERR_INVALID  # stride < row bytes
    # neighbourhood operators refuse a destination that overlaps the source (rejected before any HIP call; fake addresses)
    base = 1 << 20
    assert lib.ipx_box3_u8(None, base, base, 1, 8, 8, 1, 8, 64, 8, 64, None, None) == L.IPX_ERR_INVALID
    assert b"overlaps" in lib.ipx_last_error()
    assert lib.ipx_sobel_u8(None, base, base + 16, 1, 8, 8, 1, 8, 64, 8, 64, None, None) == L.IPX_ERR_INVALID      # shifted by two rows
    assert lib.ipx_gaussian_blur_u8(None, base, base + 63, 1, 8, 8, 1, 8, 64, 8, 64, 5, None, None) == L.IPX_ERR_INVALID
    assert lib.ipx_median_blur_u8(None, base, base + 8, 1, 8, 8, 8, 64, 8, 64, 3) == L.IPX_ERR_INVALID
    assert lib.ipx_lee_filter_u8(None, base, base + 32, 2, 8, 8, 8, 64, 8, 64, 5) == L.IPX_ERR_INVALID
    se = (ctypes.c_uint8 * 9)(*([1] * 9))
    assert lib.ipx_morph_u8(None, base, base + 4, 1, 8, 8, 8, 64, 8, 64, 0, ctypes.addressof(se), 3, 3) == L.IPX_ERR_INVALID
    assert lib.ipx_sobel_lut_u8(None, base, base + 4096, base + 4096 + 8, 1, 8, 8, 8, 64, 8, 64, base + 65536, 0) == L.IPX_ERR_INVALID
    assert lib.ipx_gaussian_blur_gray_u8(None, base, base + 4096, base + 100, 1, 8, 8, 24, 192, 24, 192, 8, 64, 5) == L.IPX_ERR_INVALID
    with pytest.raises(L.IpxError):
        L.call("ipx_equalize_lut_u8", None, None, 1, None)
    # a histogram workspace: null, misaligned, or shorter than its tickets
    assert lib.ipx_hist_workspace_init(None, None, 1 << 20) == L.IPX_ERR_INVALID
    assert lib.ipx_hist_workspace_init(None, 0x1001, 1 << 20) == L.IPX_ERR_INVALID
    assert lib.ipx_hist_workspace_init(None, 0x1000, 1024) == L.IPX_ERR_INVALID
    assert lib.ipx_hist_workspace_bytes(1, 16, 16) >= (4096 + 128) * 4


def test_cpu_tensor_is_refused():
    import torch
    import ipx.ops as ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.apply_lut(torch.zeros((4, 4), dtype=torch.uint8), np.arange(256, dtype=np.uint8))

"""ctypes binding of libipx.so (include/ipx.h).  Thin: argument types, error -> exception.

There is NO fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IPX_LIB", os.path.join(_HERE, "..", "lib", "libipx.so"))

IPX_OK, IPX_ERR_INVALID, IPX_ERR_HIP, IPX_ERR_UNSUPPORTED = 0, -1, -2, -3
MAX_GAUSS_K = 255
MAX_MULTI_LUT = 8
NOALLOC_SCRATCH_BYTES = 1536 * 1024


class IpxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libipx error {code}: {msg}")
        self.code = code


_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
_IMG = [_vp, _vp, _vp, _i, _i, _i]           # stream, src, dst, n, h, w
_STR = [_i64, _i64, _i64, _i64]              # src_row, src_img, dst_row, dst_img

# name -> (restype, argtypes); every symbol include/ipx.h declares
SIGNATURES = {
    "ipx_version": (_i, []),
    "ipx_last_error": (C.c_char_p, []),
    "ipx_check_device": (_i, []),
    "ipx_bgr2gray_u8": (_i, _IMG + [_i] + _STR),
    "ipx_gray_from_u16_u8": (_i, _IMG + [_i, _i] + _STR),
    "ipx_lut_u8": (_i, _IMG + [_i] + _STR + [_vp]),
    "ipx_lut_dev_u8": (_i, _IMG + [_i] + _STR + [_vp, _i64]),
    "ipx_lut_multi_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i] + _STR + [_vp, _i64]),
    "ipx_bgr2gray_lut_u8": (_i, _IMG + [_i] + _STR + [_vp]),
    "ipx_hist_workspace_bytes": (_sz, [_i, _i, _i]),
    "ipx_hist_workspace_init": (_i, [_vp, _vp, _sz]),
    "ipx_hist_u8": (_i, [_vp, _vp, _i, _i, _i, _i64, _i64, _vp, _vp, _sz]),
    "ipx_equalize_lut_u8": (_i, [_vp, _vp, _i, _vp]),
    "ipx_log_lut_u8": (_i, [_vp, _vp, _i, _vp, _vp, _i64]),
    "ipx_equalize_hist_u8": (_i, _IMG + _STR + [_vp, _vp, _vp, _sz]),
    "ipx_equalize_hist_noalloc_u8": (_i, _IMG + _STR + [_vp, _vp]),
    "ipx_bgr2gray_equalize_u8": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _sz]),
    "ipx_laplacian_sharpen_u8": (_i, _IMG + [_i] + _STR + [_vp, _vp]),
    "ipx_box3_u8": (_i, _IMG + [_i] + _STR + [_vp, _vp]),
    "ipx_sobel_u8": (_i, _IMG + [_i] + _STR + [_vp, _vp]),
    "ipx_sobel_lut_u8": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i] + _STR + [_vp, _i64]),
    "ipx_gauss_kernel_q8": (_i, [_i, C.c_double, _vp]),
    "ipx_sep_filter_q8_u8": (_i, _IMG + [_i] + _STR + [_vp, _vp, _i, _vp, _vp]),
    "ipx_gaussian_blur_u8": (_i, _IMG + [_i] + _STR + [_i, _vp, _vp]),
    "ipx_gaussian_blur_gray_one_pass": (_i, [_i, _i, _i]),
    "ipx_conv3_lut_u8": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i] + _STR + [_vp, _i64, _vp, _vp]),
    "ipx_gaussian_sharpen_gray_u8": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i] + _STR + [_i64, _i64, _i]),
    "ipx_gaussian_blur_gray_u8": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i] + _STR + [_i64, _i64, _i]),
    "ipx_lee_filter_u8": (_i, _IMG + _STR + [_i]),
    "ipx_pairwise_sum_lut_f32": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "ipx_median_blur_u8": (_i, _IMG + _STR + [_i]),
    "ipx_morph_u8": (_i, _IMG + _STR + [_i, _vp, _i, _i]),
    "ipx_base64_encode_u8": (_i, [_vp, _vp, _i64, _vp]),
    "ipx_png_unfilter_workspace_bytes": (_sz, [_i, _i, _i]),
    "ipx_png_unfilter_u8": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _vp, _sz, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load libipx.so once; raise if it is not there (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        path = os.path.normpath(LIB_PATH)
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: build it with `make -C image-processing_amd` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != IPX_OK:
        raise IpxError(rc, load().ipx_last_error().decode("utf-8", "replace"))


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args))

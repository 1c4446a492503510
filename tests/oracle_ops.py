"""TEST-ONLY stand-in with the ipx.ops surface, backed by the CPU oracle (torch CPU tensors in/out).

Lets the host logic (contract dispatch, batch sharder, row partition with halo exchange and histogram
all-reduce) be exercised on machines without a GPU.  Never imported by the product package."""
import numpy as np
import torch

import oracle as O


def _np(t):
    return t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _with_halo(img, top, bot, r, f):
    """Filter rows of `img` given optional halo rows: build the extended image, filter, crop."""
    a = _np(img)
    parts, cut_top, cut_bot = [a], 0, 0
    if top is not None:
        parts.insert(0, _np(top)); cut_top = r
    if bot is not None:
        parts.append(_np(bot)); cut_bot = r
    ext = np.concatenate(parts, axis=0)
    # reflect at the sides that have no halo is what the oracle does on `ext`'s own borders; on the halo sides
    # the r extra rows make the interior rows exact, provided the halo rows' own outer neighbours do not matter:
    out = f(ext)
    return _t(out[cut_top: out.shape[0] - cut_bot])


class OracleOps:
    @staticmethod
    def bgr2gray(t): return _t(O.bgr2gray_u8(_np(t)))
    @staticmethod
    def gray_from_u16(t, w, channels, big_endian=True, rgb_order=True): return _t(O.gray_u8_from_u16_bytes(_np(t), w, channels, big_endian, rgb_order))
    @staticmethod
    def apply_lut(t, lut): return _t(np.asarray(lut, np.uint8)[_np(t)])
    @staticmethod
    def apply_lut_dev(t, luts): return _t(_np(luts).reshape(-1, 256)[0][_np(t)])
    @staticmethod
    def bgr2gray_lut(t, lut): return _t(np.asarray(lut, np.uint8)[O.bgr2gray_u8(_np(t))])
    @staticmethod
    def hist(t): return _t(O.hist_u8(_np(t)).astype(np.int32))
    @staticmethod
    def equalize_lut(h): return _t(O.equalize_lut_from_hist(_np(h).astype(np.int64)))
    @staticmethod
    def equalize_hist(t): return _t(O.equalize_hist_u8(_np(t)))
    @staticmethod
    def bgr2gray_equalize(t):
        g = O.bgr2gray_u8(_np(t)); return _t(g), _t(O.equalize_hist_u8(g))
    @staticmethod
    def laplacian_sharpen(t, top_halo=None, bottom_halo=None): return _with_halo(t, top_halo, bottom_halo, 1, O.laplacian_sharpen_u8)
    @staticmethod
    def box3(t, top_halo=None, bottom_halo=None): return _with_halo(t, top_halo, bottom_halo, 1, O.box3_u8)
    @staticmethod
    def sobel(t, top_halo=None, bottom_halo=None): return _with_halo(t, top_halo, bottom_halo, 1, O.sobel_u8)
    @staticmethod
    def conv3_lut(name, t, luts, top_halo=None, bottom_halo=None):
        f = {"laplacian_sharpen": O.laplacian_sharpen_u8, "box3": O.box3_u8, "sobel": O.sobel_u8}[name]
        return _with_halo(t, top_halo, bottom_halo, 1, f), _t(_np(luts).reshape(-1, 256)[0][_np(t)])
    @staticmethod
    def gaussian_blur(t, k, top_halo=None, bottom_halo=None):
        return _with_halo(t, top_halo, bottom_halo, k // 2, lambda a: O.gaussian_blur_u8(a, k))


    @staticmethod
    def lee_filter(t, window_size=5): return _t(O.lee_filter_u8(_np(t), window_size))
    @staticmethod
    def denoise_stats(t, method_name="median"): return O.denoise_stats(_np(t), method_name)
    @staticmethod
    def median_blur(t, k): return _t(O.median_blur_u8(_np(t), k))
    @staticmethod
    def ellipse_se(m): return O.ellipse_se(m)
    @staticmethod
    def morph_open(t, se): return _t(O.morph_open_u8(_np(t), se))
    @staticmethod
    def morph_close(t, se): return _t(O.morph_close_u8(_np(t), se))


class OracleBackend:
    """app.GpuBackend stand-in: numpy in, numpy out."""
    def run(self, fn, *arrays):
        out = fn(OracleOps, *[_t(a) for a in arrays])
        if isinstance(out, tuple):
            return [o.numpy() if isinstance(o, torch.Tensor) else o for o in out]
        return out.numpy() if isinstance(out, torch.Tensor) else out

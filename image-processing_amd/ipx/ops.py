"""Tensor-level operators over libipx (PyTorch-ROCm is plumbing: device memory + streams).

Every function takes uint8 CUDA tensors laid out like the reference's numpy arrays
(`H x W`, `H x W x C`) or a batch of them (`N x H x W x C`; a gray batch is `N x H x W x 1`),
launches on `torch.cuda.current_stream()` and returns a new tensor.  Names follow the reference
call sites they replace (server.py).  No CPU fallback: a CPU tensor is an error.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib


class _Img:
    """(n, h, w, c, row_stride, img_stride, ptr) view of a tensor; inner (w, c) must be dense."""

    def __init__(self, t: torch.Tensor, name: str = "src"):
        if not isinstance(t, torch.Tensor) or t.dtype != torch.uint8:
            raise TypeError(f"{name}: uint8 torch tensor expected")
        if not t.is_cuda:
            raise RuntimeError(f"{name}: tensor is on {t.device}; libipx runs on the GPU only (no CPU fallback)")
        if t.ndim == 2:
            v = t[None, :, :, None]
        elif t.ndim == 3:
            if t.shape[2] > 4:
                raise ValueError(f"{name}: 3-D tensor is read as H x W x C with C <= 4; use N x H x W x 1 for a gray batch")
            v = t[None]
        elif t.ndim == 4:
            v = t
        else:
            raise ValueError(f"{name}: expected 2-D, 3-D or 4-D tensor, got {t.ndim}-D")
        n, h, w, c = v.shape
        if n == 0 or h == 0 or w == 0:
            raise ValueError(f"{name}: empty image {tuple(t.shape)}")
        if c > 4:
            raise ValueError(f"{name}: {c} channels (at most 4 interleaved channels)")
        if (c > 1 and v.stride(3) != 1) or (w > 1 and v.stride(2) != c):
            v = v.contiguous()
        self.t = v
        self.n, self.h, self.w, self.c = n, h, w, c
        self.row = v.stride(1) if h > 1 else w * c
        self.img = v.stride(0) if n > 1 else max(self.row * h, 1)
        if self.row < w * c or (n > 1 and self.img < self.row * (h - 1) + w * c):
            self.t = v = v.contiguous()
            self.row, self.img = w * c, w * c * h
        self.ptr = v.data_ptr()
        self.ndim_in = t.ndim

    def like(self, c: Optional[int] = None) -> torch.Tensor:
        return torch.empty((self.n, self.h, self.w, self.c if c is None else c), dtype=torch.uint8, device=self.t.device)

    def restore(self, out: torch.Tensor) -> torch.Tensor:
        if self.ndim_in == 2:
            return out[0, :, :, 0]
        if self.ndim_in == 3:
            return out[0]
        return out


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """Handle of torch's current stream on the current device (the raw getter skips building a Stream object: 0.3 us
    instead of 2.6 per call, which shows when one 4K frame per launch keeps the GPU busy for only ~10 us)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class _OnDevice:
    """`with torch.cuda.device(dev)` only when `dev` is not already current (the guard costs 2 us per call)."""
    __slots__ = ("guard",)

    def __init__(self, dev: torch.device):
        idx = dev.index
        self.guard = None if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()
        return self

    def __exit__(self, *a):
        if self.guard is not None:
            return self.guard.__exit__(*a)
        return False


def _lut_np(lut) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(lut), dtype=np.uint8)
    if a.shape != (256,):
        raise ValueError("table must have 256 uint8 entries")
    return a


def _halo(t: Optional[torch.Tensor], src: _Img, r: int, name: str):
    if t is None:
        return None, None
    if src.n != 1:
        raise ValueError("halo rows are only defined for a single image")
    hv = _Img(t, name)
    if hv.h != r or hv.w != src.w or hv.c != src.c:
        raise ValueError(f"{name}: expected {r} x {src.w} x {src.c}, got {hv.h} x {hv.w} x {hv.c}")
    if hv.t.device != src.t.device:
        raise ValueError(f"{name}: on {hv.t.device}, the image is on {src.t.device}")
    if hv.h > 1 and hv.row != src.row:
        # the library reads halo rows with the SOURCE's row stride (include/ipx.h): re-lay them out that way
        buf = torch.empty((r, src.row), dtype=torch.uint8, device=src.t.device)
        buf[:, :src.w * src.c].view(r, src.w, src.c).copy_(hv.t[0])
        return buf, buf.data_ptr()
    return hv, hv.ptr


# ---- pixel-wise ----------------------------------------------------------------------------------
def bgr2gray(src: torch.Tensor) -> torch.Tensor:
    """cv2.cvtColor(img, COLOR_BGR2GRAY) (server.py:517); a 4th channel is dropped (server.py:511)."""
    s = _Img(src)
    if s.c not in (3, 4):
        raise ValueError("bgr2gray needs 3 (BGR) or 4 (BGRA) channels")
    with _OnDevice(s.t.device):
        out = s.like(1)
        _lib.call("ipx_bgr2gray_u8", _stream(), s.ptr, out.data_ptr(), s.n, s.h, s.w, s.c,
                  s.row, s.img, s.w, s.w * s.h)
    return out[0, :, :, 0] if src.ndim == 3 else out


def gray_from_u16(src: torch.Tensor, w: int, channels: int, big_endian: bool = True, rgb_order: bool = True) -> torch.Tensor:
    """The 16-bit branch of ensure_gray_uint8_from_bytes (server.py:516-523): cvtColor(BGR2GRAY) on uint16 samples, then >> 8.
    `src` holds the samples as BYTES: (h, 2 * w * channels) or (n, h, 2 * w * channels) uint8 -- the scan lines png_unfilter
    reconstructs (big-endian, R first: the defaults) or a numpy uint16 array viewed as bytes (big_endian=False).  channels 1..4
    (1, 2: only the shift; 4: alpha dropped).  Returns (h, w) or (n, h, w) uint8."""
    if not isinstance(src, torch.Tensor) or src.dtype != torch.uint8 or src.ndim not in (2, 3):
        raise TypeError("src: 2-D or 3-D uint8 tensor of sample bytes expected")
    if not src.is_cuda:
        raise RuntimeError(f"src: tensor is on {src.device}; libipx runs on the GPU only (no CPU fallback)")
    if channels not in (1, 2, 3, 4) or w <= 0 or src.shape[-1] != 2 * w * channels:
        raise ValueError(f"src rows hold {src.shape[-1]} bytes, 2 x {w} x {channels} expected")
    v = src if src.ndim == 3 else src[None]
    if v.stride(2) != 1:
        v = v.contiguous()
    n, h = v.shape[0], v.shape[1]
    if n == 0 or h == 0:
        raise ValueError(f"src: empty image {tuple(src.shape)}")
    row = v.stride(1) if h > 1 else v.shape[2]
    img = v.stride(0) if n > 1 else max(row * h, 1)
    if row < v.shape[2] or (n > 1 and img < row * (h - 1) + v.shape[2]):
        v = v.contiguous()
        row, img = v.shape[2], v.shape[2] * h
    with _OnDevice(v.device):
        out = torch.empty((n, h, w), dtype=torch.uint8, device=v.device)
        _lib.call("ipx_gray_from_u16_u8", _stream(), v.data_ptr(), out.data_ptr(), n, h, w, channels,
                  (1 if big_endian else 0) | (2 if rgb_order else 0), row, img, w, w * h)
    return out[0] if src.ndim == 2 else out


def apply_lut(src: torch.Tensor, lut) -> torch.Tensor:
    """dst = lut[src] with one 256-entry host table (negative/threshold/gamma/log/piecewise, server.py:362-388,443-467)."""
    s = _Img(src)
    tab = _lut_np(lut)
    with _OnDevice(s.t.device):
        out = s.like()
        _lib.call("ipx_lut_u8", _stream(), s.ptr, out.data_ptr(), s.n, s.h, s.w, s.c,
                  s.row, s.img, s.w * s.c, s.w * s.c * s.h, tab.ctypes.data)
    return s.restore(out)


def apply_lut_dev(src: torch.Tensor, luts: torch.Tensor) -> torch.Tensor:
    """Per-image device tables: luts is (256,) shared or (N, 256)."""
    s = _Img(src)
    if luts.dtype != torch.uint8 or not luts.is_cuda or luts.shape[-1] != 256:
        raise ValueError("luts: uint8 CUDA tensor of shape (256,) or (N, 256) expected")
    luts = luts.contiguous()
    stride = 0 if luts.ndim == 1 else 256
    if luts.ndim == 2 and luts.shape[0] != s.n:
        raise ValueError("one table per image expected")
    with _OnDevice(s.t.device):
        out = s.like()
        _lib.call("ipx_lut_dev_u8", _stream(), s.ptr, out.data_ptr(), s.n, s.h, s.w, s.c,
                  s.row, s.img, s.w * s.c, s.w * s.c * s.h, luts.data_ptr(), stride)
    return s.restore(out)


def apply_luts_dev(src: torch.Tensor, luts: torch.Tensor) -> list:
    """k tables applied to ONE read of the source (ipx_lut_multi_u8): luts is (k, 256) shared or (N, k, 256) per image,
    uint8 on the source's device, k <= 8; returns the k outputs in table order."""
    import ctypes
    s = _Img(src)
    if luts.dtype != torch.uint8 or luts.device != s.t.device or luts.shape[-1] != 256 or luts.ndim not in (2, 3):
        raise ValueError("luts: uint8 tensor of shape (k, 256) or (N, k, 256) on the source's device expected")
    luts = luts.contiguous()
    k = luts.shape[-2]
    if not 1 <= k <= _lib.MAX_MULTI_LUT:
        raise ValueError(f"1..{_lib.MAX_MULTI_LUT} tables expected")
    if luts.ndim == 3 and luts.shape[0] != s.n:
        raise ValueError("one set of tables per image expected")
    with _OnDevice(s.t.device):
        outs = [s.like() for _ in range(k)]
        ptrs = (ctypes.c_void_p * k)(*[o.data_ptr() for o in outs])
        _lib.call("ipx_lut_multi_u8", _stream(), s.ptr, ctypes.cast(ptrs, ctypes.c_void_p), k, s.n, s.h, s.w, s.c,
                  s.row, s.img, s.w * s.c, s.w * s.c * s.h, luts.data_ptr(), 0 if luts.ndim == 2 else 256 * k)
    return [s.restore(o) for o in outs]


def bgr2gray_lut(src: torch.Tensor, lut) -> torch.Tensor:
    """Fused cvtColor(BGR2GRAY) + table (one 4 B/px pass instead of 4 + 2)."""
    s = _Img(src)
    if s.c not in (3, 4):
        raise ValueError("bgr2gray_lut needs 3 (BGR) or 4 (BGRA) channels")
    tab = _lut_np(lut)
    with _OnDevice(s.t.device):
        out = s.like(1)
        _lib.call("ipx_bgr2gray_lut_u8", _stream(), s.ptr, out.data_ptr(), s.n, s.h, s.w, s.c,
                  s.row, s.img, s.w, s.w * s.h, tab.ctypes.data)
    return out[0, :, :, 0] if src.ndim == 3 else out


# ---- histogram -----------------------------------------------------------------------------------
def _gray(src: torch.Tensor) -> _Img:
    s = _Img(src)
    if s.c != 1:
        raise ValueError("single-channel (gray) image expected")
    return s


_ws_cache: dict = {}            # (device index, stream handle) -> initialised workspace tensor
_ws_in_graphs: list = []        # workspaces a captured graph refers to: kept for the life of the process


def _workspace(s: _Img) -> torch.Tensor:
    """Scratch for the histogram kernels: one buffer per (device, stream), grown on demand and prepared once with
    ipx_hist_workspace_init (the kernels keep its tickets at zero themselves, so there is no memset and no allocation per
    call).  Calls on one stream are ordered, which is what sharing the buffer needs.  Under graph capture the stream's
    cached buffer is used when an eager call on that stream made one before (the graph then holds kernel launches only, and
    the buffer is never released); otherwise the buffer comes from the capture's own pool and is initialised inside the graph."""
    lib = _lib.load()
    nbytes = lib.ipx_hist_workspace_bytes(s.n, s.h, s.w)
    dev = s.t.device
    capturing = torch.cuda.is_current_stream_capturing()
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream())
    ws = _ws_cache.get(key)
    if ws is not None and ws.numel() >= nbytes:
        if capturing and not any(w is ws for w in _ws_in_graphs):
            _ws_in_graphs.append(ws)
        return ws
    ws = torch.empty((max(nbytes, 2 << 20),), dtype=torch.uint8, device=dev)
    _lib.call("ipx_hist_workspace_init", _stream(), ws.data_ptr(), ws.numel())
    if not capturing:
        if len(_ws_cache) >= 64:                         # streams come and go: drop the lot rather than track them
            _ws_cache.clear()
        _ws_cache[key] = ws
    return ws


def hist(src: torch.Tensor) -> torch.Tensor:
    """256-bin histogram per image -> int32 tensor (N, 256) (or (256,) for a single H x W image).
    Counts are uint32 in the library; exposed as int32 because image sizes are capped at 2^31-1 pixels."""
    s = _gray(src)
    with _OnDevice(s.t.device):
        out = torch.empty((s.n, 256), dtype=torch.int32, device=s.t.device)
        ws = _workspace(s)
        _lib.call("ipx_hist_u8", _stream(), s.ptr, s.n, s.h, s.w, s.row, s.img, out.data_ptr(), ws.data_ptr(), ws.numel())
    return out[0] if src.ndim <= 3 else out


def equalize_lut(hists: torch.Tensor) -> torch.Tensor:
    """cv2.equalizeHist's table from histograms (N,256) int32 -> (N,256) uint8."""
    if hists.dtype != torch.int32 or not hists.is_cuda or hists.shape[-1] != 256:
        raise ValueError("hists: int32 CUDA tensor (..., 256) expected")
    h2 = hists.reshape(-1, 256).contiguous()
    with _OnDevice(h2.device):
        out = torch.empty((h2.shape[0], 256), dtype=torch.uint8, device=h2.device)
        _lib.call("ipx_equalize_lut_u8", _stream(), h2.data_ptr(), h2.shape[0], out.data_ptr())
    return out.reshape(hists.shape)


def log_lut(hists: torch.Tensor, curve: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mode=log's tables (server.py:385-388) from histograms (N,256) int32 and the float32 curve log1p(c * level/255)
    (256 values on the device, luts.log_curve) -> (N,256) uint8.  `out` may be a row-strided (N,256) view, e.g. one table
    slot of the N x k x 256 table sets apply_luts_dev takes."""
    if hists.dtype != torch.int32 or not hists.is_cuda or hists.shape[-1] != 256:
        raise ValueError("hists: int32 CUDA tensor (..., 256) expected")
    if curve.dtype != torch.float32 or curve.device != hists.device or curve.numel() != 256 or not curve.is_contiguous():
        raise ValueError("curve: 256 contiguous float32 values on the histograms' device expected")
    h2 = hists.reshape(-1, 256).contiguous()
    n = h2.shape[0]
    with _OnDevice(h2.device):
        if out is None:
            out = torch.empty((n, 256), dtype=torch.uint8, device=h2.device)
        if out.dtype != torch.uint8 or out.device != h2.device or tuple(out.shape) != (n, 256) or out.stride(1) != 1 or (n > 1 and out.stride(0) < 256):
            raise ValueError("out: (N,256) uint8 view with unit column stride and row stride >= 256 expected")
        _lib.call("ipx_log_lut_u8", _stream(), h2.data_ptr(), n, curve.data_ptr(), out.data_ptr(), out.stride(0) if n > 1 else 256)
    return out


def equalize_hist(src: torch.Tensor, return_aux: bool = False):
    """cv2.equalizeHist(img) (server.py:246)."""
    s = _gray(src)
    with _OnDevice(s.t.device):
        out = s.like()
        hst = lut = None
        ws = _workspace(s)
        if return_aux:
            hst = torch.empty((s.n, 256), dtype=torch.int32, device=s.t.device)
            lut = torch.empty((s.n, 256), dtype=torch.uint8, device=s.t.device)
        _lib.call("ipx_equalize_hist_u8", _stream(), s.ptr, out.data_ptr(), s.n, s.h, s.w,
                  s.row, s.img, s.w, s.w * s.h,
                  hst.data_ptr() if return_aux else None, lut.data_ptr() if return_aux else None,
                  ws.data_ptr(), ws.numel())
    res = s.restore(out)
    return (res, hst, lut) if return_aux else res


def equalize_hist_noalloc(src: torch.Tensor, out: torch.Tensor, hist_out: Optional[torch.Tensor] = None,
                          lut_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """cv2.equalizeHist through the entry point that allocates nothing (ipx_equalize_hist_noalloc_u8): the caller
    passes the output (and optional (N,256) int32 histogram / uint8 table) tensors; the scratch is the library's
    built-in per-GPU array.  Calls that overlap on one GPU race -- keep them on one stream."""
    s, d = _gray(src), _gray(out)
    if (d.n, d.h, d.w) != (s.n, s.h, s.w) or d.t.data_ptr() != (out.data_ptr() if out.ndim else 0):
        raise ValueError("out: a writable tensor of the source's shape expected (no copy may be made of it)")
    for t, dt, nm in ((hist_out, torch.int32, "hist_out"), (lut_out, torch.uint8, "lut_out")):
        if t is not None and (t.dtype != dt or not t.is_cuda or not t.is_contiguous() or t.numel() != s.n * 256):
            raise ValueError(f"{nm}: contiguous {dt} CUDA tensor with N*256 entries expected")
    with _OnDevice(s.t.device):
        _lib.call("ipx_equalize_hist_noalloc_u8", _stream(), s.ptr, d.ptr, s.n, s.h, s.w, s.row, s.img, d.row, d.img,
                  hist_out.data_ptr() if hist_out is not None else None,
                  lut_out.data_ptr() if lut_out is not None else None)
    return out


def bgr2gray_equalize(src: torch.Tensor, return_aux: bool = False):
    """server.py:245-246 on a colour image in one pass over the source: returns (gray, equalized[, hist, lut])."""
    s = _Img(src)
    if s.c not in (3, 4):
        raise ValueError("bgr2gray_equalize needs 3 (BGR) or 4 (BGRA) channels")
    with _OnDevice(s.t.device):
        gray, out = s.like(1), s.like(1)
        ws = _workspace(s)
        hst = lut = None
        if return_aux:
            hst = torch.empty((s.n, 256), dtype=torch.int32, device=s.t.device)
            lut = torch.empty((s.n, 256), dtype=torch.uint8, device=s.t.device)
        _lib.call("ipx_bgr2gray_equalize_u8", _stream(), s.ptr, gray.data_ptr(), out.data_ptr(), s.n, s.h, s.w, s.c,
                  s.row, s.img, s.w, s.w * s.h, s.w, s.w * s.h,
                  hst.data_ptr() if return_aux else None, lut.data_ptr() if return_aux else None, ws.data_ptr(), ws.numel())
    if src.ndim == 3:
        gray, out = gray[0, :, :, 0], out[0, :, :, 0]
    return (gray, out, hst, lut) if return_aux else (gray, out)


def hist_tables(src: torch.Tensor):
    """Histograms and cv2.equalizeHist tables of a gray batch without the table pass (ipx_equalize_hist_u8 with dst = NULL):
    returns (hist int32 (N,256), tables uint8 (N,256)); (256,) each for a single H x W image."""
    s = _gray(src)
    with _OnDevice(s.t.device):
        h = torch.empty((s.n, 256), dtype=torch.int32, device=s.t.device)
        lut = torch.empty((s.n, 256), dtype=torch.uint8, device=s.t.device)
        ws = _workspace(s)
        _lib.call("ipx_equalize_hist_u8", _stream(), s.ptr, None, s.n, s.h, s.w, s.row, s.img, 0, 0,
                  h.data_ptr(), lut.data_ptr(), ws.data_ptr(), ws.numel())
    return (h[0], lut[0]) if src.ndim == 2 else (h, lut)


def bgr2gray_hist(src: torch.Tensor):
    """cvtColor(BGR2GRAY) + per-image histogram + equalizeHist TABLE in one pass over the colour source, without applying
    the table (ipx_bgr2gray_equalize_u8 with dst = NULL): returns (gray, hist (N x 256 int32), lut (N x 256 uint8))."""
    s = _Img(src)
    if s.c not in (3, 4):
        raise ValueError("bgr2gray_hist needs 3 (BGR) or 4 (BGRA) channels")
    with _OnDevice(s.t.device):
        gray = s.like(1)
        ws = _workspace(s)
        hst = torch.empty((s.n, 256), dtype=torch.int32, device=s.t.device)
        lut = torch.empty((s.n, 256), dtype=torch.uint8, device=s.t.device)
        _lib.call("ipx_bgr2gray_equalize_u8", _stream(), s.ptr, gray.data_ptr(), None, s.n, s.h, s.w, s.c,
                  s.row, s.img, s.w, s.w * s.h, s.w, s.w * s.h, hst.data_ptr(), lut.data_ptr(), ws.data_ptr(), ws.numel())
    if src.ndim == 3:
        gray = gray[0, :, :, 0]
    return gray, hst, lut


# ---- convolutions --------------------------------------------------------------------------------
def _conv3(name: str, src: torch.Tensor, top: Optional[torch.Tensor], bottom: Optional[torch.Tensor]) -> torch.Tensor:
    s = _Img(src)
    tv, tp = _halo(top, s, 1, "top_halo")
    bv, bp = _halo(bottom, s, 1, "bottom_halo")
    with _OnDevice(s.t.device):
        out = s.like()
        _lib.call(name, _stream(), s.ptr, out.data_ptr(), s.n, s.h, s.w, s.c,
                  s.row, s.img, s.w * s.c, s.w * s.c * s.h, tp, bp)
    return s.restore(out)


def laplacian_sharpen(src, top_halo=None, bottom_halo=None):
    """clip(f64(c) - cv2.Laplacian(c, CV_64F), 0, 255) per channel (server.py:266-276)."""
    return _conv3("ipx_laplacian_sharpen_u8", src, top_halo, bottom_halo)


def box3(src, top_halo=None, bottom_halo=None):
    """3x3 normalised box blur (BASELINE.json configs[0]; not in the reference)."""
    return _conv3("ipx_box3_u8", src, top_halo, bottom_halo)


def sobel(src, top_halo=None, bottom_halo=None):
    """min(255, |gx| + |gy|) (BASELINE.json configs[1]; not in the reference)."""
    return _conv3("ipx_sobel_u8", src, top_halo, bottom_halo)


def sobel_lut(src: torch.Tensor, luts: torch.Tensor):
    """sobel(src) and apply_lut_dev(src, luts) of one gray batch in one pass over it (ipx_sobel_lut_u8): returns
    (sobel, table[src]).  luts: (256,) shared or (N,256) per image, uint8 on the device."""
    s = _gray(src)
    if luts.dtype != torch.uint8 or luts.device != s.t.device or luts.shape[-1] != 256:
        raise ValueError("luts: uint8 tensor (..., 256) on the image's device expected")
    l2 = luts.reshape(-1, 256).contiguous()
    if l2.shape[0] not in (1, s.n):
        raise ValueError(f"luts: {l2.shape[0]} tables for {s.n} images")
    with _OnDevice(s.t.device):
        a, b = s.like(), s.like()
        _lib.call("ipx_sobel_lut_u8", _stream(), s.ptr, a.data_ptr(), b.data_ptr(), s.n, s.h, s.w,
                  s.row, s.img, s.w, s.w * s.h, l2.data_ptr(), 256 if l2.shape[0] > 1 else 0)
    return s.restore(a), s.restore(b)


_CONV3_OPS = {"laplacian_sharpen": 0, "box3": 1, "sobel": 2}


def conv3_lut(name: str, src: torch.Tensor, luts: torch.Tensor, top_halo=None, bottom_halo=None):
    """A 3x3 operator (name in 'laplacian_sharpen', 'box3', 'sobel') and apply_lut_dev(src, luts) of one gray batch in one pass over it
    (ipx_conv3_lut_u8), with the optional halo rows of a row-partitioned image: returns (filtered, table[src])."""
    if name not in _CONV3_OPS:
        raise ValueError(f"conv3_lut: unknown operator {name!r}")
    s = _gray(src)
    if luts.dtype != torch.uint8 or luts.device != s.t.device or luts.shape[-1] != 256:
        raise ValueError("luts: uint8 tensor (..., 256) on the image's device expected")
    l2 = luts.reshape(-1, 256).contiguous()
    if l2.shape[0] not in (1, s.n):
        raise ValueError(f"luts: {l2.shape[0]} tables for {s.n} images")
    tv, tp = _halo(top_halo, s, 1, "top_halo")
    bv, bp = _halo(bottom_halo, s, 1, "bottom_halo")
    with _OnDevice(s.t.device):
        a, b = s.like(), s.like()
        _lib.call("ipx_conv3_lut_u8", _stream(), _CONV3_OPS[name], s.ptr, a.data_ptr(), b.data_ptr(), s.n, s.h, s.w,
                  s.row, s.img, s.w, s.w * s.h, l2.data_ptr(), 256 if l2.shape[0] > 1 else 0, tp, bp)
    return s.restore(a), s.restore(b)


def gauss_kernel_q8(k: int, sigma: float = 0.0) -> np.ndarray:
    """Q0.8 taps cv2.GaussianBlur uses on uint8 (library helper, host side)."""
    if k < 1 or k % 2 == 0 or k > _lib.MAX_GAUSS_K:
        raise ValueError(f"k={k} must be odd and in 1..{_lib.MAX_GAUSS_K}")
    out = np.zeros(k, np.uint16)
    _lib.call("ipx_gauss_kernel_q8", int(k), C.c_double(sigma), out.ctypes.data)
    return out


def sep_filter_q8(src, kx: Sequence[int], ky: Sequence[int], top_halo=None, bottom_halo=None) -> torch.Tensor:
    s = _Img(src)
    kx = np.ascontiguousarray(kx, np.uint16)
    ky = np.ascontiguousarray(ky, np.uint16)
    if kx.shape != ky.shape or kx.ndim != 1:
        raise ValueError("kx and ky must have the same odd length")
    r = len(kx) // 2
    tv, tp = _halo(top_halo, s, r, "top_halo") if r else (None, None)
    bv, bp = _halo(bottom_halo, s, r, "bottom_halo") if r else (None, None)
    with _OnDevice(s.t.device):
        out = s.like()
        _lib.call("ipx_sep_filter_q8_u8", _stream(), s.ptr, out.data_ptr(), s.n, s.h, s.w, s.c,
                  s.row, s.img, s.w * s.c, s.w * s.c * s.h, kx.ctypes.data, ky.ctypes.data, len(kx), tp, bp)
    return s.restore(out)


def gaussian_blur(src, k: int, top_halo=None, bottom_halo=None) -> torch.Tensor:
    """cv2.GaussianBlur(img, (k, k), 0) on uint8 (server.py:428)."""
    s = _Img(src)
    r = k // 2
    tv, tp = _halo(top_halo, s, r, "top_halo") if r else (None, None)
    bv, bp = _halo(bottom_halo, s, r, "bottom_halo") if r else (None, None)
    with _OnDevice(s.t.device):
        out = s.like()
        _lib.call("ipx_gaussian_blur_u8", _stream(), s.ptr, out.data_ptr(), s.n, s.h, s.w, s.c,
                  s.row, s.img, s.w * s.c, s.w * s.c * s.h, int(k), tp, bp)
    return s.restore(out)


# ---- Lee filter, denoise_manual statistics -------------------------------
def gaussian_blur_gray_one_pass(h: int, w: int, k: int) -> bool:
    """True if gaussian_blur_gray serves h x w BGR images with one launch (one read of the colour batch)."""
    return bool(_lib.load().ipx_gaussian_blur_gray_one_pass(int(h), int(w), int(k)))


def gaussian_blur_gray(src: torch.Tensor, k: int):
    """gaussian_blur(src, k) and bgr2gray(src) of one BGR batch in one pass over it (ipx_gaussian_blur_gray_u8): returns
    (blurred colour batch, gray of the unblurred source)."""
    s = _Img(src)
    if s.c != 3:
        raise ValueError("gaussian_blur_gray needs 3 (BGR) channels")
    with _OnDevice(s.t.device):
        out, gray = s.like(), s.like(1)
        _lib.call("ipx_gaussian_blur_gray_u8", _stream(), s.ptr, out.data_ptr(), gray.data_ptr(), s.n, s.h, s.w,
                  s.row, s.img, s.w * 3, s.w * 3 * s.h, s.w, s.w * s.h, int(k))
    return s.restore(out), (gray[0, :, :, 0] if src.ndim == 3 else gray)


def gaussian_sharpen_gray(src: torch.Tensor, k: int = 5):
    """gaussian_blur(src, k), laplacian_sharpen(src) and bgr2gray(src) of one BGR batch in one pass over it
    (ipx_gaussian_sharpen_gray_u8): returns (blurred, sharpened, gray)."""
    s = _Img(src)
    if s.c != 3:
        raise ValueError("gaussian_sharpen_gray needs 3 (BGR) channels")
    with _OnDevice(s.t.device):
        out, sharp, gray = s.like(), s.like(), s.like(1)
        _lib.call("ipx_gaussian_sharpen_gray_u8", _stream(), s.ptr, out.data_ptr(), sharp.data_ptr(), gray.data_ptr(), s.n, s.h, s.w,
                  s.row, s.img, s.w * 3, s.w * 3 * s.h, s.w, s.w * s.h, int(k))
    return s.restore(out), s.restore(sharp), (gray[0, :, :, 0] if src.ndim == 3 else gray)


def lee_filter(src: torch.Tensor, window_size: int = 5) -> torch.Tensor:
    """lee_filter(img, window_size) (server.py:711-732): float32 window mean/variance blend on interior pixels."""
    s = _gray(src)
    with _OnDevice(s.t.device):
        out = s.like()
        _lib.call("ipx_lee_filter_u8", _stream(), s.ptr, out.data_ptr(), s.n, s.h, s.w,
                  s.row, s.img, s.w, s.w * s.h, int(window_size))
    return s.restore(out)


def pairwise_sum_lut(src: torch.Tensor, table) -> np.float32:
    """np.add.reduce(table[img.flatten()].astype(np.float32)) in numpy's own order of float32 additions (blocking)."""
    s = _gray(src)
    t = s.t if (s.row == s.w and (s.n == 1 or s.img == s.w * s.h)) else s.t.contiguous()
    tab = np.ascontiguousarray(table, np.float32)
    if tab.shape != (256,):
        raise ValueError("table must have 256 float32 entries")
    res = np.zeros(1, np.float32)
    with _OnDevice(t.device):
        _lib.call("ipx_pairwise_sum_lut_f32", _stream(), t.data_ptr(), int(t.numel()), tab.ctypes.data, res.ctypes.data)
    return res[0]


def denoise_stats(src: torch.Tensor, method_name: str = "median") -> dict:
    """The histogram analysis of mode=denoise_manual (server.py:190-211) for ONE gray image, bit-identical to the
    reference: counts of 0 / 255 from the GPU histogram; np.mean / np.var / np.std of the float32 pixel array are
    float32 sums in numpy's fixed order, formed on the GPU (ipx_pairwise_sum_lut_f32); the scalar steps around them are
    the reference's own numpy expressions on float32 scalars / 256-level tables."""
    if src.ndim != 2:
        raise ValueError("denoise_stats: one H x W gray image expected")
    h = hist(src).to("cpu").numpy().astype(np.int64)
    n = int(src.numel())
    total = float(n)
    extreme_ratio = float((int(h[0]) + int(h[255])) / total)                       # server.py:193-195
    lv = np.arange(256, dtype=np.uint8).astype(np.float32)
    mean32 = np.float32(pairwise_sum_lut(src, lv) / n)                             # np.mean(flat): sum / count in float32
    mean_val = float(mean32)
    dev = lv - mean32                                                              # np.var: x = arr - arrmean; x = x * x
    var32 = np.float32(pairwise_sum_lut(src, dev * dev) / n)
    var = float(var32)
    cv_val = float(np.sqrt(var32) / (mean_val + 1e-12))                            # np.std(flat) / (mean_val + 1e-12)
    if extreme_ratio > 0.02:
        suggested = "median"
    elif cv_val > 0.28:
        suggested = "lee"
    elif 100 < var < 2000:
        suggested = "bilateral"
    else:
        suggested = method_name
    return {"extreme_ratio": extreme_ratio, "mean": mean_val, "variance": var, "cv": cv_val,
            "suggested_method": suggested}


def stats_from_hist(h: np.ndarray, method_name: str = "median") -> dict:
    """server.py:190-211 from a 256-bin histogram alone (float64 on exact counts): what a row-partitioned image can
    compute after its histogram all-reduce.  Agrees with denoise_stats to float32 rounding, not bit for bit."""
    h = np.asarray(h, np.int64)
    total = float(h.sum())
    extreme_ratio = float((int(h[0]) + int(h[255])) / total)
    lv = np.arange(256, dtype=np.float64)
    mean_val = float((h * lv).sum() / total)
    var = float((h * (lv - mean_val) ** 2).sum() / total)
    cv_val = float(np.sqrt(var) / (mean_val + 1e-12))
    if extreme_ratio > 0.02:
        suggested = "median"
    elif cv_val > 0.28:
        suggested = "lee"
    elif 100 < var < 2000:
        suggested = "bilateral"
    else:
        suggested = method_name
    return {"extreme_ratio": extreme_ratio, "mean": mean_val, "variance": var, "cv": cv_val,
            "suggested_method": suggested}


# ---- median and morphology of the `median` pipeline -----------------
def median_blur(src: torch.Tensor, k: int) -> torch.Tensor:
    """cv2.medianBlur(img, k) on gray uint8 (server.py:215, 430): BORDER_REPLICATE, exact median."""
    s = _gray(src)
    with _OnDevice(s.t.device):
        out = s.like()
        _lib.call("ipx_median_blur_u8", _stream(), s.ptr, out.data_ptr(), s.n, s.h, s.w, s.row, s.img, s.w, s.w * s.h, int(k))
    return s.restore(out)


def ellipse_se(m: int) -> np.ndarray:
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (m, m)) (server.py:432), OpenCV's row-extent formula."""
    r = c = m // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    se = np.zeros((m, m), np.uint8)
    for i in range(m):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))       # saturate_cast<int> = round half to even
            se[i, max(c - dx, 0):min(c + dx + 1, m)] = 1
    return se


def _morph(src: torch.Tensor, se, op: int) -> torch.Tensor:
    s = _gray(src)
    se = np.ascontiguousarray(se, np.uint8)
    if se.ndim != 2:
        raise ValueError("structuring element must be 2-D")
    with _OnDevice(s.t.device):
        out = s.like()
        _lib.call("ipx_morph_u8", _stream(), s.ptr, out.data_ptr(), s.n, s.h, s.w, s.row, s.img, s.w, s.w * s.h,
                  op, se.ctypes.data, se.shape[0], se.shape[1])
    return s.restore(out)


def erode(src, se): return _morph(src, se, 0)
def dilate(src, se): return _morph(src, se, 1)
def morph_open(src, se): return dilate(erode(src, se), se)      # cv2.morphologyEx(MORPH_OPEN), server.py:433
def morph_close(src, se): return erode(dilate(src, se), se)     # cv2.morphologyEx(MORPH_CLOSE), server.py:434


# ---- decode staging ---------------------------------------------------------------------------------
def png_unfilter(raw: torch.Tensor, h: int, w: int, bytes_per_pixel: int):
    """PNG reconstruction (un-filtering, PNG specification section 9) of inflated scan lines on the GPU: `raw` is a flat
    uint8 CUDA tensor holding h rows of [filter type][w * bytes_per_pixel filtered bytes].  Returns (pixels, status):
    pixels (h, w * bytes_per_pixel) uint8 in file order (16-bit samples big-endian); status a 1-element int32 tensor that is 0
    when the pixels are valid (read it after the stream is synchronised; see ipx_png_unfilter_u8)."""
    if not isinstance(raw, torch.Tensor) or raw.dtype != torch.uint8 or not raw.is_cuda or not raw.is_contiguous():
        raise TypeError("raw: contiguous uint8 CUDA tensor expected (no CPU fallback)")
    stride = 1 + w * bytes_per_pixel
    if h <= 0 or w <= 0 or raw.numel() < h * stride:
        raise ValueError(f"raw holds {raw.numel()} bytes, {h} rows of {stride} expected")
    with _OnDevice(raw.device):
        out = torch.empty((h, w * bytes_per_pixel), dtype=torch.uint8, device=raw.device)
        status = torch.empty((1,), dtype=torch.int32, device=raw.device)
        nws = _lib.load().ipx_png_unfilter_workspace_bytes(h, w, bytes_per_pixel)
        ws = torch.empty((nws,), dtype=torch.uint8, device=raw.device)
        _lib.call("ipx_png_unfilter_u8", _stream(), raw.data_ptr(), stride, out.data_ptr(), w * bytes_per_pixel,
                  h, w, bytes_per_pixel, ws.data_ptr(), ws.numel(), status.data_ptr())
    return out, status


def base64_encode(data: torch.Tensor) -> torch.Tensor:
    """base64.b64encode (server.py:536) of a flat uint8 CUDA tensor -> uint8 tensor of 4 * ceil(n / 3) ASCII characters."""
    if not isinstance(data, torch.Tensor) or data.dtype != torch.uint8 or not data.is_cuda or data.ndim != 1 or not data.is_contiguous():
        raise TypeError("data: flat contiguous uint8 CUDA tensor expected (no CPU fallback)")
    n = data.numel()
    with _OnDevice(data.device):
        out = torch.empty((4 * ((n + 2) // 3),), dtype=torch.uint8, device=data.device)
        if n:
            _lib.call("ipx_base64_encode_u8", _stream(), data.data_ptr(), n, out.data_ptr())
    return out

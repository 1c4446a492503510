"""256-entry tables for the reference's pixel-wise transforms (host side, numpy).

Every pixel-wise mode of server.py is a function of the pixel value alone (log additionally of the
image's set of present levels), so the GPU only has to *apply* a table; the table itself is built by
evaluating the reference's own numpy expressions on the 256 possible values -- same dtype, same
order of operations -- which is what makes the result bit-identical to the whole-image evaluation.
This is host logic of the product (it runs per request, 256 elements); it is not a CPU fallback for
the image pass.
"""
from __future__ import annotations

import numpy as np

_LEVELS = np.arange(256, dtype=np.uint8)


def negative() -> np.ndarray:
    """server.py:376  `255 - img`."""
    return (255 - _LEVELS).astype(np.uint8)


def threshold(T: int) -> np.ndarray:
    """server.py:380  cv2.threshold(img, T, 255, THRESH_BINARY): v > T ? 255 : 0 (T<0 -> all 255, T>=255 -> all 0)."""
    return np.where(_LEVELS.astype(np.int64) > int(np.floor(T)), 255, 0).astype(np.uint8)


def gamma(gamma_val: float) -> np.ndarray:
    """server.py:370-372."""
    r = _LEVELS.astype(np.float32) / 255.0
    s = np.power(r, gamma_val)
    return np.clip((s * 255.0), 0, 255).astype(np.uint8)


def log_curve(c_val: float) -> np.ndarray:
    """log1p(c * r) of server.py:385-386 on the 256 levels (float32, increasing in the level)."""
    r = _LEVELS.astype(np.float32) / 255.0
    return np.log1p(c_val * r)


def log(c_val: float, present: np.ndarray) -> np.ndarray:
    """server.py:385-388.  `s.max()` runs over the image, i.e. over the levels that occur in it:
    `present` is a 256-vector (histogram or boolean) marking them."""
    present = np.asarray(present).reshape(256) > 0
    if not present.any():
        raise ValueError("empty image")
    s = log_curve(c_val)
    with np.errstate(all="ignore"):
        s /= (s[present].max() + 1e-12)
        return (s * 255.0).astype(np.uint8)


def piecewise_clamp(r1: int, s1: int, r2: int, s2: int):
    """server.py:454-457."""
    r1 = max(1, min(254, r1))
    r2 = max(r1 + 1, min(255, r2))
    s1 = int(np.clip(s1, 0, 255))
    s2 = int(np.clip(s2, 0, 255))
    return r1, s1, r2, s2


def piecewise(r1: int, s1: int, r2: int, s2: int) -> np.ndarray:
    """server.py:459-467 (parameters already clamped by piecewise_clamp)."""
    x = _LEVELS.astype(np.float32)
    y = np.zeros_like(x, dtype=np.float32)
    idx1 = x <= r1
    y[idx1] = (s1 / float(r1)) * x[idx1]
    idx2 = (x > r1) & (x <= r2)
    y[idx2] = s1 + (s2 - s1) * (x[idx2] - r1) / float(r2 - r1)
    idx3 = x > r2
    with np.errstate(divide="ignore", invalid="ignore"):
        y[idx3] = s2 + (255.0 - s2) * (x[idx3] - r2) / float(255 - r2)
    return np.clip(np.round(y), 0, 255).astype(np.uint8)

"""The BASELINE.json workloads as sequences of ipx.ops calls (shared by bench.py, scripts/ and the GPU config tests).

Every step is one reference call site (server.py line cited at the op); nothing here computes pixels itself.
"""
from __future__ import annotations

import numpy as np

from . import luts

FULL_PIPELINE_TABLES = (("negative", luts.negative), ("threshold", lambda: luts.threshold(128)),
                        ("gamma", lambda: luts.gamma(0.6)),
                        ("piecewise", lambda: luts.piecewise(*luts.piecewise_clamp(70, 10, 180, 245))))


_DEV_CONST = {}


def _dev_const(key, device, make):
    """Small constant tensors (tables, curves) kept on each device: uploading them per call would put a blocking pageable
    copy into the middle of an otherwise asynchronous pipeline."""
    k = (key, str(device))
    if k not in _DEV_CONST:
        _DEV_CONST[k] = make().to(device)
    return _DEV_CONST[k]


def log_tables_dev(hists, c: float = 2.0):
    """mode=log tables (server.py:385-388) for a batch, built ON THE DEVICE from the images' histograms (N x 256 counts):
    `s.max()` is per image and `s` increases with the level, so the maximum is `s` at the highest level present; then the
    reference's two float32 operations per level, `s / (max + 1e-12)` and `* 255`, and the uint8 truncation -- the same
    IEEE operations torch performs elementwise (checked against luts.log for every possible highest level in
    tests/test_gpu_parity.py).  No host round trip: the pipeline stays asynchronous."""
    import torch
    dev = hists.device
    s = _dev_const(("log_curve", float(c)), dev, lambda: torch.from_numpy(luts.log_curve(c)))   # float32 (256,), numpy's log1p
    levels = _dev_const("levels", dev, lambda: torch.arange(256))
    eps = _dev_const("eps", dev, lambda: torch.tensor(1e-12, dtype=torch.float32))
    top = ((hists.reshape(-1, 256) > 0) * levels).amax(dim=1)                         # highest level present, per image
    d = s[top] + eps
    return ((s[None, :] / d[:, None]) * 255.0).to(torch.uint8)                        # N x 256


def log_transform(ops, gray, c: float = 2.0):
    """mode=log (server.py:383-388) on a batch: each image gets its OWN table (log_tables_dev), applied in one launch."""
    return ops.apply_lut_dev(gray, log_tables_dev(ops.hist(gray), c))


def full_pipeline(ops, bgr) -> dict:
    """BASELINE.json configs[2] "full filter pipeline" on a batch of BGR frames (N x H x W x 3): every pixel-wise mode
    of the reference on the gray image (server.py:360 then :376, :380, :370-372, :385-388, :459-467), histogram
    equalization (:245-246), colour Laplacian sharpening (:263-277) and the 5x5 Gaussian (:428, per channel).
    Returns the nine output batches by mode name (plus 'gray').
    The six table modes (equalizeHist's table pass among them) read the gray batch ONCE (ipx_lut_multi_u8: 1 + 6 bytes per
    pixel instead of 12); the log table is per image (`s.max()`, server.py:387), built on the device from the histogram the
    gray conversion already produced."""
    import torch
    gray, hists, eq_lut = ops.bgr2gray_hist(bgr)                # one pass over the colour source: gray8, histograms, equalize tables
    out = {"gray": gray}
    n = hists.reshape(-1, 256).shape[0]
    shared = _dev_const("full_pipeline_tables", gray.device,
                        lambda: torch.from_numpy(np.stack([make() for _, make in FULL_PIPELINE_TABLES]).astype(np.uint8)))
    tabs = torch.cat([shared[None].expand(n, -1, -1), log_tables_dev(hists, 2.0)[:, None, :], eq_lut.reshape(n, 1, 256)],
                     dim=1).contiguous()                            # N x 6 x 256
    res = ops.apply_luts_dev(gray if gray.ndim == 4 else gray[None, :, :, None], tabs)
    if gray.ndim != 4:
        res = [r[0, :, :, 0] for r in res]
    for (name, _), r in zip(FULL_PIPELINE_TABLES, res):
        out[name] = r
    out["log"] = res[len(FULL_PIPELINE_TABLES)]
    out["histogram_eq"] = res[len(FULL_PIPELINE_TABLES) + 1]
    out["sharpening"] = ops.laplacian_sharpen(bgr)
    out["gaussian5"] = ops.gaussian_blur(bgr, 5)
    return out


def configs1_step(ops, bgr):
    """BASELINE.json configs[1]: 5x5 Gaussian (BGR) + hist-eq (incl. BGR->gray) + Sobel on the gray image.  cvtColor +
    equalizeHist run as the fused entry point's two halves with Sobel between them: Sobel then reads the gray batch right
    after it was written (same kernels and results as bgr2gray_equalize followed by sobel, 2.5 % less time per step)."""
    g5 = ops.gaussian_blur(bgr, 5)
    gray, _, lut = ops.bgr2gray_hist(bgr)
    sb = ops.sobel(gray)
    return g5, ops.apply_lut_dev(gray, lut), sb

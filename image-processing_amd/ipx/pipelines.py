"""The BASELINE.json workloads as sequences of ipx.ops calls (shared by bench.py, scripts/ and the GPU config tests).

Every step is one reference call site (server.py line cited at the op); nothing here computes pixels itself.
"""
from __future__ import annotations

import numpy as np

from . import luts

FULL_PIPELINE_TABLES = (("negative", luts.negative), ("threshold", lambda: luts.threshold(128)),
                        ("gamma", lambda: luts.gamma(0.6)),
                        ("piecewise", lambda: luts.piecewise(*luts.piecewise_clamp(70, 10, 180, 245))))


def log_transform(ops, gray, c: float = 2.0):
    """mode=log (server.py:383-388) on a batch: `s.max()` is per image, so each image gets its OWN table, built on the
    host from that image's device histogram (one small D2H of N x 256 counts) and applied with per-image device
    tables in one launch."""
    import torch
    hists = ops.hist(gray)
    present = hists.reshape(-1, 256).cpu().numpy()
    tabs = np.stack([luts.log(c, p) for p in present])
    if tabs.shape[0] == 1:
        return ops.apply_lut(gray, tabs[0])
    return ops.apply_lut_dev(gray, torch.from_numpy(tabs).to(gray.device, non_blocking=True))


def full_pipeline(ops, bgr) -> dict:
    """BASELINE.json configs[2] "full filter pipeline" on a batch of BGR frames (N x H x W x 3): every pixel-wise mode
    of the reference on the gray image (server.py:360 then :376, :380, :370-372, :385-388, :459-467), histogram
    equalization (:245-246), colour Laplacian sharpening (:263-277) and the 5x5 Gaussian (:428, per channel).
    Returns the nine output batches by mode name (plus 'gray')."""
    gray, eq = ops.bgr2gray_equalize(bgr)                      # one pass over the colour source for gray8 + histogram_eq
    out = {"gray": gray, "histogram_eq": eq}
    for name, make in FULL_PIPELINE_TABLES:
        out[name] = ops.apply_lut(gray, make())
    out["log"] = log_transform(ops, gray, 2.0)
    out["sharpening"] = ops.laplacian_sharpen(bgr)
    out["gaussian5"] = ops.gaussian_blur(bgr, 5)
    return out


def configs1_step(ops, bgr):
    """BASELINE.json configs[1]: 5x5 Gaussian (BGR) + hist-eq (incl. BGR->gray) + Sobel on the gray image."""
    g5 = ops.gaussian_blur(bgr, 5)
    gray, eq = ops.bgr2gray_equalize(bgr)
    return g5, eq, ops.sobel(gray)

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


def log_tables_dev(hists, c: float = 2.0, out=None, ops=None):
    """mode=log tables (server.py:385-388) for a batch, built ON THE DEVICE from the images' histograms (N x 256 counts) by
    ipx_log_lut_u8: `s.max()` is per image and `s` increases with the level, so the maximum is `s` at the highest level
    present; then the reference's two float32 operations per level, `s / (max + 1e-12)` and `* 255`, and the uint8
    truncation (checked against luts.log for every possible highest level in tests/test_gpu_parity.py).  No host round trip:
    the pipeline stays asynchronous; one launch (round 2 spelled it with nine torch kernels)."""
    import torch
    if ops is None:
        from . import ops
    s = _dev_const(("log_curve", float(c)), hists.device, lambda: torch.from_numpy(luts.log_curve(c)))   # float32 (256,), numpy's log1p
    return ops.log_lut(hists, s, out)


def log_transform(ops, gray, c: float = 2.0):
    """mode=log (server.py:383-388) on a batch: each image gets its OWN table (log_tables_dev), applied in one launch."""
    return ops.apply_lut_dev(gray, log_tables_dev(ops.hist(gray), c, ops=ops))


def full_pipeline(ops, bgr) -> dict:
    """BASELINE.json configs[2] "full filter pipeline" on a batch of BGR frames (N x H x W x 3): every pixel-wise mode
    of the reference on the gray image (server.py:360 then :376, :380, :370-372, :385-388, :459-467), histogram
    equalization (:245-246), colour Laplacian sharpening (:263-277) and the 5x5 Gaussian (:428, per channel).
    Returns the nine output batches by mode name (plus 'gray').
    The six table modes (equalizeHist's table pass among them) read the gray batch ONCE (ipx_lut_multi_u8: 1 + 6 bytes per
    pixel instead of 12); the log table is per image (`s.max()`, server.py:387), built on the device from the histogram the
    gray conversion already produced."""
    import torch
    one_pass = getattr(ops, "gaussian_blur_gray_one_pass", None)
    if one_pass is not None and one_pass(bgr.shape[-3], bgr.shape[-2], 5):
        # the Gaussian, the sharpened and the gray batch from ONE read of the colour source (ipx_gaussian_sharpen_gray_u8: 10 bytes
        # per pixel instead of 6 + 6 + 4), then histograms + equalize tables from the gray batch, whose loads leave it in the
        # Infinity Cache for the six-table pass.  128 x 1080p, same-process A/B (scripts/dev/cfg3ab.py): 1110 us per batch with three
        # reads of the colour batch, 1016 with the Gaussian and the gray batch sharing one, 860 with all three.
        g5, sharp, gray = ops.gaussian_sharpen_gray(bgr, 5)
        hists, eq_lut = ops.hist_tables(gray)
    else:
        g5 = sharp = None
        gray, hists, eq_lut = ops.bgr2gray_hist(bgr)            # one pass over the colour source: gray8, histograms, equalize tables
    out = {"gray": gray}
    n = hists.reshape(-1, 256).shape[0]
    shared = _dev_const("full_pipeline_tables", gray.device,
                        lambda: torch.from_numpy(np.stack([make() for _, make in FULL_PIPELINE_TABLES]).astype(np.uint8)))
    k = len(FULL_PIPELINE_TABLES)
    tabs = torch.empty((n, k + 2, 256), dtype=torch.uint8, device=gray.device)      # N x 6 x 256
    tabs[:, :k] = shared
    log_tables_dev(hists, 2.0, out=tabs[:, k], ops=ops)
    tabs[:, k + 1] = eq_lut.reshape(n, 256)
    res = ops.apply_luts_dev(gray if gray.ndim == 4 else gray[None, :, :, None], tabs)
    if gray.ndim != 4:
        res = [r[0, :, :, 0] for r in res]
    for (name, _), r in zip(FULL_PIPELINE_TABLES, res):
        out[name] = r
    out["log"] = res[len(FULL_PIPELINE_TABLES)]
    out["histogram_eq"] = res[len(FULL_PIPELINE_TABLES) + 1]
    out["sharpening"] = sharp if sharp is not None else ops.laplacian_sharpen(bgr)
    out["gaussian5"] = g5 if g5 is not None else ops.gaussian_blur(bgr, 5)
    return out


def configs1_step(ops, bgr):
    """BASELINE.json configs[1]: 5x5 Gaussian (BGR) + hist-eq (incl. BGR->gray) + Sobel on the gray image, every input read
    once: the Gaussian and cvtColor share one read of the colour batch (ipx_gaussian_blur_gray_u8), the gray batch's
    histograms and equalize tables come from one read of it, Sobel and equalizeHist's table pass share the next
    (ipx_sobel_lut_u8).  Same pixels as gaussian_blur, bgr2gray_equalize and sobel called one after the other; 11 instead of
    14 bytes per pixel (scripts/dev/gaussgray.py, sobellut.py).  Frames whose rows the one-pass kernel does not take (rows shorter than
    4 KiB or not whole 16-byte lanes) keep the Gaussian alone and cvtColor + histogram + tables as one launch (13 bytes per pixel)."""
    one_pass = getattr(ops, "gaussian_blur_gray_one_pass", None)
    if one_pass is not None and one_pass(bgr.shape[-3], bgr.shape[-2], 5):
        g5, gray = ops.gaussian_blur_gray(bgr, 5)
        _, lut = ops.hist_tables(gray)
    else:
        g5 = ops.gaussian_blur(bgr, 5)
        gray, _, lut = ops.bgr2gray_hist(bgr)
    sb, eq = ops.sobel_lut(gray, lut)
    return g5, eq, sb


def configs1_one_image(ops, bgr, side_stream=None):
    """BASELINE.json configs[1] taken literally -- ONE image per launch, what three single /api/process requests do: the
    5x5 Gaussian (server.py:428), the fused cvtColor + equalizeHist call (:245-246) and Sobel on the gray image.  Returns
    (gaussian, gray, equalized, sobel).  With `side_stream` the Gaussian, which shares nothing with the other two but
    its input, is enqueued there (forked from and joined to the current stream): one frame leaves the chip mostly idle
    during each kernel's ramp and tail, and under graph capture the fork becomes two branches that overlap."""
    import torch
    one_pass = getattr(ops, "gaussian_blur_gray_one_pass", None)
    if side_stream is None and one_pass is not None and one_pass(bgr.shape[-3], bgr.shape[-2], 5):
        g5, gray = ops.gaussian_blur_gray(bgr, 5)                # the Gaussian and the gray frame from one read of the colour frame
        _, lut = ops.hist_tables(gray)                           # histogram + table (one launch for one 4K frame)
    elif side_stream is None:
        g5 = ops.gaussian_blur(bgr, 5)
        gray, _, lut = ops.bgr2gray_hist(bgr)
    else:                                                        # forked form: the Gaussian alone on the side stream
        cur = torch.cuda.current_stream()
        side_stream.wait_stream(cur)
        with torch.cuda.stream(side_stream):
            g5 = ops.gaussian_blur(bgr, 5)
        gray, _, lut = ops.bgr2gray_hist(bgr)
    sb, eq = ops.sobel_lut(gray, lut)                            # Sobel + the table pass on one read of the gray frame
    if side_stream is not None:
        torch.cuda.current_stream().wait_stream(side_stream)
    return g5, gray, eq, sb


class GraphedStep:
    """A fixed sequence of ipx.ops calls on inputs of fixed shapes, captured ONCE into a HIP graph and replayed per request
    (VERDICT r02 item 5).  One 4K frame keeps the GPU busy for ~10-20 us per kernel, the same order as the time the host
    needs to enqueue a launch through Python and ctypes; a replay enqueues the whole step with one call.

        step = GraphedStep(lambda x: pipelines.configs1_one_image(ops, x), frame)
        outs = step(new_frame)            # copies new_frame into step.inputs[0], replays; outs are the SAME tensors each time
        step.inputs[0].copy_(pinned, non_blocking=True); outs = step()      # or upload straight into the static input

    Inside the capture every output and temporary comes from the graph's private memory pool and the histogram workspace is
    the capture stream's cached one, initialised during the eager warm-up on that stream -- the graph holds kernel launches
    only.  Outputs are overwritten by the next replay: consume (or copy) them first, on the stream the replay ran on."""

    def __init__(self, fn, *example, warmup: int = 3):
        import torch
        if not example or not all(isinstance(e, torch.Tensor) and e.is_cuda for e in example):
            raise ValueError("GraphedStep: CUDA tensors expected as example inputs")
        self.fn = fn
        dev = example[0].device
        self.inputs = [e.clone() for e in example]
        self.stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for _ in range(max(1, warmup)):            # loads code objects, fills the per-stream workspace cache, warms the allocator
                fn(*self.inputs)
        self.stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.outputs = fn(*self.inputs)
        cur.wait_stream(self.stream)

    def __call__(self, *xs):
        if xs:
            if len(xs) != len(self.inputs):
                raise ValueError(f"GraphedStep: {len(self.inputs)} inputs expected, got {len(xs)}")
            for dst, x in zip(self.inputs, xs):
                if x is not dst:
                    if x.shape != dst.shape or x.dtype != dst.dtype:
                        raise ValueError(f"GraphedStep was captured for {tuple(dst.shape)} {dst.dtype}, got {tuple(x.shape)} {x.dtype}")
                    dst.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.outputs

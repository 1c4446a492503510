#!/usr/bin/env python3
"""Dev (IPX_LIB -> the -DIPX_DEV library): launch-geometry sweeps for ONE 4K frame per launch (configs[1] literal).
Each figure: median over 40 event-timed launches, each preceded by a small unrelated kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (2160, 3840, 3), dtype=torch.uint8, device=dev)
gray = ops.bgr2gray(bgr)
mark = torch.zeros(1 << 16, device=dev)
REPS = 20
def t(fn, n=15):
    """GPU time per launch: REPS launches captured into one graph (geometry knobs are read at capture), median replay / REPS.
    (Event-timed eager launches of one frame measure the host: ~18 us per call through Python and ctypes.)"""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): fn()
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(REPS): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        mark.add_(1.0)
        a.record(); g.replay(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ts[len(ts) // 2] * 1e3 / REPS, 2)
def sweep(name, fn, var, values, extra=None):
    res = []
    for v in values:
        if v is None: os.environ.pop(var, None)
        else: os.environ[var] = str(v)
        for k, x in (extra or {}).items(): os.environ[k] = str(x)
        res.append((v, t(fn)))
    os.environ.pop(var, None)
    for k in (extra or {}): os.environ.pop(k, None)
    print(name, var, extra or "", res, flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "gauss"):
    sweep("gaussian5 bgr", lambda: ops.gaussian_blur(bgr, 5), "IPX_BAND_ROWS", [None, 2, 4, 6, 8, 12])
    sweep("gaussian5 bgr", lambda: ops.gaussian_blur(bgr, 5), "IPX_BAND_ROWS", [2, 4, 6, 8], {"IPX_BAND_LANES": 128})
    sweep("gaussian5 bgr", lambda: ops.gaussian_blur(bgr, 5), "IPX_BAND_ROWS", [2, 4, 6, 8], {"IPX_BAND_LANES": 192})
    sweep("gaussian5 bgr walker", lambda: ops.gaussian_blur(bgr, 5), "IPX_SEG_ROWS", [6, 12, 24, 48], {"IPX_BAND_ROWS": 0})
    sweep("gaussian3 bgr", lambda: ops.gaussian_blur(bgr, 3), "IPX_BAND_ROWS", [None, 2, 4, 8])
if which in ("all", "sobel"):
    sweep("sobel gray", lambda: ops.sobel(gray), "IPX_SEG_ROWS", [None, 4, 8, 12, 16, 24])
    sweep("sobel gray wpb1", lambda: ops.sobel(gray), "IPX_SEG_ROWS", [4, 8, 16], {"IPX_WPB": 1})
    sweep("sobel gray wpb2", lambda: ops.sobel(gray), "IPX_SEG_ROWS", [4, 8, 16], {"IPX_WPB": 2})
    sweep("sobel gray band", lambda: ops.sobel(gray), "IPX_BAND3_ROWS", [0, 4, 8, 16])
    sweep("sharpen bgr", lambda: ops.laplacian_sharpen(bgr), "IPX_SEG_ROWS", [None, 4, 8, 16])
    sweep("sharpen bgr band", lambda: ops.laplacian_sharpen(bgr), "IPX_BAND3_ROWS", [0, 4, 8, 16])
    sweep("box3 gray", lambda: ops.box3(gray), "IPX_SEG_ROWS", [None, 4, 8, 16])
if which in ("all", "hist"):
    for slots in (512, 1024):
        sweep("bgr2gray_equalize", lambda: ops.bgr2gray_equalize(bgr), "IPX_HIST_GRAIN", [65536, 32768, 16384], {"IPX_HIST_SLOTS": slots})
        sweep("bgr2gray_hist", lambda: ops.bgr2gray_hist(bgr), "IPX_HIST_GRAIN", [65536, 32768, 16384], {"IPX_HIST_SLOTS": slots})
        sweep("equalize gray", lambda: ops.equalize_hist(gray), "IPX_HIST_GRAIN", [65536, 32768, 16384], {"IPX_HIST_SLOTS": slots})
    sweep("bgr2gray_equalize no ticket", lambda: ops.bgr2gray_equalize(bgr), "IPX_HIST_GRAIN", [65536, 32768, 16384], {"IPX_HIST_NOTICKET": 1})
    lut_t = torch.arange(255, -1, -1, dtype=torch.uint8, device=dev)
    sweep("lut gray", lambda: ops.apply_lut_dev(gray, lut_t), "IPX_NONE", [None])
    sweep("bgr2gray", lambda: ops.bgr2gray(bgr), "IPX_NONE", [None])

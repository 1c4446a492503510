#!/usr/bin/env python3
"""Dev: the 5-tap Gaussian (walker: IPX_BAND_ROWS=0; band tiles: rows per tile) on 24 x 4K BGR after different predecessors:
a small gray kernel, a big BGR writer (sharpen), itself.  Events around the call; chip warmed up.  -DIPX_DEV library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
N, H, W = int(os.environ.get("N", "24")), int(os.environ.get("H", "2160")), int(os.environ.get("W", "3840"))
bgr = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device=dev)
gray = ops.bgr2gray(bgr)
def pipe_time(pre, fn, reps=10):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps + 3):
        pre()
        if i >= 3: ev[i - 3][0].record()
        fn()
        if i >= 3: ev[i - 3][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ts[len(ts) // 2] * 1e3, 1)
for _ in range(200):
    ops.laplacian_sharpen(bgr)
torch.cuda.synchronize()
g5 = lambda: ops.gaussian_blur(bgr, 5)
pres = {"after sobel(gray)": lambda: ops.sobel(gray), "after lut(bgr)": lambda: ops.apply_lut(bgr, list(range(256))),
        "after sharpen(bgr)": lambda: ops.laplacian_sharpen(bgr), "after itself": g5}
for rows in [0] + [int(x) for x in os.environ.get("ROWS", "8,10,12").split(",")] + [0]:
    os.environ["IPX_BAND_ROWS"] = str(rows)
    print(f"rows={rows}:", {k: pipe_time(p, g5) for k, p in pres.items()}, flush=True)

#!/usr/bin/env python3
"""Dev A/B of the bench step WITHOUT events between its kernels (what a pipeline without instrumentation sees): walker
(IPX_BAND_ROWS=0) against band tiles for the 5-tap Gaussian; also the Gaussian alone back to back.  -DIPX_DEV library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device=dev)
def step():
    ops.gaussian_blur(bgr, 5)
    gray, eq = ops.bgr2gray_equalize(bgr)
    ops.sobel(gray)
def total(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n * 1e3, 1)
for _ in range(100): step()
for rows in os.environ.get("ROWS", "0,12,0,12,8,11,14").split(","):
    os.environ["IPX_BAND_ROWS"] = rows
    print(f"rows={rows}: step {total(step)} us; gaussian alone back to back {total(lambda: ops.gaussian_blur(bgr, 5))} us", flush=True)

#!/usr/bin/env python3
"""Dev: event-free pairs / triples of kernels, walker vs band for the 5-tap Gaussian (IPX_BAND_ROWS) behind a band-tile sharpen,
on 24 x 4K and 128 x 1080p BGR.  -DIPX_DEV library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
def total(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n * 1e3, 1)
for shape in ((24, 2160, 3840, 3), (128, 1080, 1920, 3)):
    bgr = torch.randint(0, 256, shape, dtype=torch.uint8, device=dev)
    for _ in range(50): ops.laplacian_sharpen(bgr)
    for rows in ("0", "8", "0", "8"):
        os.environ["IPX_BAND_ROWS"] = rows
        pair = lambda: (ops.laplacian_sharpen(bgr), ops.gaussian_blur(bgr, 5))
        print(shape, f"g5 rows={rows}: sharpen+g5 {total(pair)} us; g5 alone {total(lambda: ops.gaussian_blur(bgr, 5))}; sharpen alone {total(lambda: ops.laplacian_sharpen(bgr))}", flush=True)
    del bgr

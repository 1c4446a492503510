#!/usr/bin/env python3
"""Dev: the band-tile kernels on 24 x 4K BGR (B, H, W from the environment) back to back, for the tile geometry the environment
selects (IPX_BAND_ROWS_ALL, IPX_BAND_ROUND; dev library)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
B, H, W = int(os.environ.get("B", "24")), int(os.environ.get("H", "2160")), int(os.environ.get("W", "3840"))
bgr = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device=dev)
def run(fn, n=40):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
res = {"gauss5": run(lambda: ops.gaussian_blur(bgr, 5)), "gauss3": run(lambda: ops.gaussian_blur(bgr, 3)),
       "sharpen": run(lambda: ops.laplacian_sharpen(bgr)), "box3": run(lambda: ops.box3(bgr)), "sobel": run(lambda: ops.sobel(bgr)),
       "gauss5+gray": run(lambda: ops.gaussian_blur_gray(bgr, 5)), "gauss3+gray": run(lambda: ops.gaussian_blur_gray(bgr, 3)),
       "gauss5+sharpen+gray": run(lambda: ops.gaussian_sharpen_gray(bgr, 5))}
print({k: os.environ.get(k) for k in ("IPX_BAND_ROWS_ALL", "IPX_BAND_ROUND")}, {k: round(v, 1) for k, v in res.items()}, flush=True)

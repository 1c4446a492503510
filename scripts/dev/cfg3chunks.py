#!/usr/bin/env python3
"""Dev: BASELINE configs[2] (128 x 1080p) as ONE pass per operator over the whole batch, or in sub-batches of CHUNK frames whose colour
bytes (CHUNK x 6.2 MB) the three readers of the colour source can find in the 256 MiB Infinity Cache."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
N = 128
bgr = torch.randint(0, 256, (N, 1080, 1920, 3), dtype=torch.uint8, device=dev)
tabs = torch.randint(0, 256, (N, 6, 256), dtype=torch.uint8, device=dev)
def run_chunks(chunk, order):
    outs = []
    for i in range(0, N, chunk):
        b = bgr[i:i + chunk]
        if order == 0:
            gray, h, lut = ops.bgr2gray_hist(b); res = ops.apply_luts_dev(gray, tabs[i:i + chunk]); sh = ops.laplacian_sharpen(b); g5 = ops.gaussian_blur(b, 5)
        else:
            g5 = ops.gaussian_blur(b, 5); sh = ops.laplacian_sharpen(b); gray, h, lut = ops.bgr2gray_hist(b); res = ops.apply_luts_dev(gray, tabs[i:i + chunk])
        outs.append((g5, sh, res))
    return outs
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for chunk in (128, 64, 32, 16, 8):
    print(f"chunk {chunk}: order gray-first {t(lambda: run_chunks(chunk, 0)):.1f} us   gaussian-first {t(lambda: run_chunks(chunk, 1)):.1f} us", flush=True)

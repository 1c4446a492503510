#!/usr/bin/env python3
"""Dev A/B on 128 x 1080p BGR (BASELINE configs[2]'s colour half): (A) 5-tap Gaussian (strip walker) | cvtColor + histogram + tables, or
(B) Gaussian + gray in one launch on band tiles of 192 lanes (IPX_GRAY_ONEPASS_MIN=150, dev library) | histogram + tables of the gray batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (int(os.environ.get("B", "128")), 1080, 1920, 3), dtype=torch.uint8, device=dev)
def a():
    g5 = ops.gaussian_blur(bgr, 5); gray, _, lut = ops.bgr2gray_hist(bgr); return g5, gray, lut
def b():
    g5, gray = ops.gaussian_blur_gray(bgr, 5); _, lut = ops.hist_tables(gray); return g5, gray, lut
print("one pass:", ops.gaussian_blur_gray_one_pass(1080, 1920, 5), "equal:", [bool((x == y).all()) for x, y in zip(a(), b())], flush=True)
def run(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for _ in range(20): a()
for rep in range(3):
    print(f"A {run(a):.1f} us   B {run(b):.1f} us", flush=True)
print(f"gaussian {run(lambda: ops.gaussian_blur(bgr, 5)):.1f}  gray+hist {run(lambda: ops.bgr2gray_hist(bgr)):.1f}  gaussian+gray {run(lambda: ops.gaussian_blur_gray(bgr, 5)):.1f}", flush=True)

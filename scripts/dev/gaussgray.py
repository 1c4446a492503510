#!/usr/bin/env python3
"""Dev A/B: the configs[1] step as (A) Gaussian | BGR->gray + histogram + tables | Sobel + table pass, or (B) Gaussian AND gray from
one read of the colour batch (ipx_gaussian_blur_gray_u8) | histogram of the gray batch + tables | Sobel + table pass.
24 x 4K BGR resident, event-free steps, alternating blocks of 50 steps; then the kernels alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "24"))
H, W = int(os.environ.get("H", "2160")), int(os.environ.get("W", "3840"))
bgr = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device=dev)
def step_a():
    g5 = ops.gaussian_blur(bgr, 5)
    gray, _, lut = ops.bgr2gray_hist(bgr)
    sb, eq = ops.sobel_lut(gray, lut)
    return g5, gray, sb, eq
def step_b():
    g5, gray = ops.gaussian_blur_gray(bgr, 5)
    _, lut = ops.hist_tables(gray)
    sb, eq = ops.sobel_lut(gray, lut)
    return g5, gray, sb, eq
ra, rb = step_a(), step_b()
torch.cuda.synchronize()
print("equal:", [bool((x == y).all()) for x, y in zip(ra, rb)], flush=True)
def run(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for _ in range(100): step_a()
for rep in range(4):
    print(f"step A {run(step_a):.1f} us   step B (gauss+gray fused) {run(step_b):.1f} us", flush=True)
gray = ops.bgr2gray(bgr)
print(f"gaussian alone {run(lambda: ops.gaussian_blur(bgr, 5)):.1f}  gaussian+gray {run(lambda: ops.gaussian_blur_gray(bgr, 5)):.1f}  "
      f"bgr2gray_hist {run(lambda: ops.bgr2gray_hist(bgr)):.1f}  hist(gray)+tables {run(lambda: ops.hist_tables(gray)):.1f}", flush=True)

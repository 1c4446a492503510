#!/usr/bin/env python3
"""Dev A/B: the configs[1] step with Sobel and the equalize table pass as two launches (A) or as ipx_sobel_lut_u8 (B);
24 x 4K BGR resident, event-free steps, alternating blocks of 50 steps; plus the two / the one launch alone after the gray + histogram pass."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "24"))
bgr = torch.randint(0, 256, (B, 2160, 3840, 3), dtype=torch.uint8, device=dev)
def step_a():
    g5 = ops.gaussian_blur(bgr, 5)
    gray, _, lut = ops.bgr2gray_hist(bgr)
    sb = ops.sobel(gray)
    eq = ops.apply_lut_dev(gray, lut)
    return g5, sb, eq
def step_b():
    g5 = ops.gaussian_blur(bgr, 5)
    gray, _, lut = ops.bgr2gray_hist(bgr)
    sb, eq = ops.sobel_lut(gray, lut)
    return g5, sb, eq
ra, rb = step_a(), step_b()
torch.cuda.synchronize()
print("equal:", all(bool((x == y).all()) for x, y in zip(ra, rb)), flush=True)
def run(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for _ in range(100): step_a()
for rep in range(4):
    print(f"step two launches {run(step_a):.1f} us   fused {run(step_b):.1f} us", flush=True)
gray, _, lut = ops.bgr2gray_hist(bgr)
def tail_a(): ops.bgr2gray_hist(bgr); ops.sobel(gray); ops.apply_lut_dev(gray, lut)
def tail_b(): ops.bgr2gray_hist(bgr); ops.sobel_lut(gray, lut)
def head(): ops.bgr2gray_hist(bgr)
for rep in range(3):
    h = run(head); print(f"after gray+hist: sobel + table {run(tail_a) - h:.1f} us   fused {run(tail_b) - h:.1f} us", flush=True)

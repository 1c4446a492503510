#!/usr/bin/env python3
"""Dev: the bench step with the equalize split into its two halves and Sobel moved between them (gray is then read by Sobel
right after it was written), against the shipped order.  Event-free totals, 24 x 4K BGR."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device=dev)
def step_a():
    ops.gaussian_blur(bgr, 5)
    gray, eq = ops.bgr2gray_equalize(bgr)
    ops.sobel(gray)
def step_b():
    ops.gaussian_blur(bgr, 5)
    gray, hst, lut = ops.bgr2gray_hist(bgr)
    ops.sobel(gray)
    ops.apply_lut_dev(gray, lut)
def step_c():
    gray, hst, lut = ops.bgr2gray_hist(bgr)
    ops.sobel(gray)
    ops.apply_lut_dev(gray, lut)
    ops.gaussian_blur(bgr, 5)
def step_d():
    gray, hst, lut = ops.bgr2gray_hist(bgr)
    ops.apply_lut_dev(gray, lut)
    ops.sobel(gray)
    ops.gaussian_blur(bgr, 5)
def total(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n * 1e3, 1)
for _ in range(100): step_a()
for rep in range(2):
    print({"shipped": total(step_a), "gauss, gray+hist, sobel, table": total(step_b), "gray+hist, sobel, table, gauss": total(step_c),
           "gray+hist, table, sobel, gauss": total(step_d)}, flush=True)

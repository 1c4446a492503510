#!/usr/bin/env python3
"""A/B of the packed-16-bit Gaussian (sep_pow2_kernel) against the general small kernel: run with IPX_LIB=lib_dev/libipx.so,
once plain and once with IPX_NO_POW2=1.  Back-to-back launches between HIP events, 24 x 4K."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device=dev)
gray = ops.bgr2gray(bgr)
ob, og = torch.empty_like(bgr), torch.empty_like(gray)
def t(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
res = {}
for k in (3, 5):
    res[f"g{k}_bgr"] = round(t(lambda: ops.gaussian_blur(bgr, k)), 1)
    res[f"g{k}_gray"] = round(t(lambda: ops.gaussian_blur(gray, k)), 1)
print(os.environ.get("IPX_NO_POW2", "pow2"), res)

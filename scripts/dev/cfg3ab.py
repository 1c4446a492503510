#!/usr/bin/env python3
"""Dev A/B, same process: ipx.pipelines.full_pipeline on 128 x 1080p BGR with the one-pass Gaussian + gray route (B) and with the
two-read route (A: the shape query patched to say no), alternating blocks of 10 batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
import ipx.pipelines as P
dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (128, 1080, 1920, 3), dtype=torch.uint8, device=dev)
real = ops.gaussian_blur_gray_one_pass
def run(n=10):
    out = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = None
        out = P.full_pipeline(ops, bgr)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for route in ("B", "A"):
    ops.gaussian_blur_gray_one_pass = real if route == "B" else (lambda h, w, k: False)
    run(5)
for rep in range(4):
    r = {}
    for route in ("A", "B"):
        ops.gaussian_blur_gray_one_pass = real if route == "B" else (lambda h, w, k: False)
        r[route] = run()
    print(f"two reads {r['A']:.1f} us   one pass {r['B']:.1f} us", flush=True)

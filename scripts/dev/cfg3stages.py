#!/usr/bin/env python3
"""Dev: BASELINE configs[2] (128 x 1080p) stage by stage, HIP events between the calls of ipx.pipelines.full_pipeline's sequence, for the
route the library picks (IPX_GRAY_ONEPASS_MIN=999 on the dev library: the two-read route)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
import ipx.pipelines as P
dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (128, 1080, 1920, 3), dtype=torch.uint8, device=dev)
tabs = torch.randint(0, 256, (128, 6, 256), dtype=torch.uint8, device=dev)
one = ops.gaussian_blur_gray_one_pass(1080, 1920, 5)
def stages():
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    if one:
        g5, gray = ops.gaussian_blur_gray(bgr, 5); ev[1].record()
        h, lut = ops.hist_tables(gray); ev[2].record()
    else:
        gray, h, lut = ops.bgr2gray_hist(bgr); ev[1].record(); ev[2].record()
    res = ops.apply_luts_dev(gray, tabs); ev[3].record()
    sh = ops.laplacian_sharpen(bgr); ev[4].record()
    if not one:
        g5 = ops.gaussian_blur(bgr, 5)
    ev[5].record()
    return ev
for _ in range(10): stages()
acc = [0.0] * 5
for _ in range(20):
    ev = stages(); torch.cuda.synchronize()
    for i in range(5): acc[i] += ev[i].elapsed_time(ev[i + 1]) * 1e3 / 20
print("one pass" if one else "two reads", [round(a, 1) for a in acc], "sum", round(sum(acc), 1), flush=True)
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): P.full_pipeline(ops, bgr)
torch.cuda.synchronize(); print("full_pipeline", round((time.perf_counter() - t0) / 20 * 1e6, 1), "us", flush=True)

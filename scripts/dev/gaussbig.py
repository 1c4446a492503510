#!/usr/bin/env python3
"""Dev: cv2.GaussianBlur with more than 31 taps (the generic two-pass path) on 24 x 4K gray and on one 8K BGR frame, ms per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
gray = torch.randint(0, 256, (24, 2160, 3840, 1), dtype=torch.uint8, device="cuda:0")
for k in [int(x) for x in os.environ.get("KS", "31,33,45,63,127,255").split(",")]:
    ops.gaussian_blur(gray, k); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): ops.gaussian_blur(gray, k)
    b.record(); torch.cuda.synchronize()
    print(f"gaussian {k} taps: {a.elapsed_time(b) / 3:.3f} ms per 24 x 4K gray", flush=True)

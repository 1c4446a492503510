#!/usr/bin/env python3
"""Dev: erode / dilate with OpenCV's elliptical structuring elements of 7 ... 31 on 24 x 4K gray, ms per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
gray = torch.randint(0, 256, (24, 2160, 3840, 1), dtype=torch.uint8, device="cuda:0")
for m in [int(x) for x in os.environ.get("SIZES", "5,7,9,15,31").split(",")]:
    se = ops.ellipse_se(m)
    for name, fn in (("erode", ops.erode), ("dilate", ops.dilate)):
        fn(gray, se); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): fn(gray, se)
        b.record(); torch.cuda.synchronize()
        print(f"{name} {m}x{m} ellipse ({int(se.sum())} members): {a.elapsed_time(b) / 3:.3f} ms per 24 x 4K", flush=True)

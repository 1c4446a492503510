#!/usr/bin/env python3
"""Dev A/B: 3 x 3 operators on 24 x 4K GRAY (240 lanes per row) as walker (IPX_BAND3_ROWS=0) and on band tiles, in the bench's
position: after bgr2gray_equalize, events around the call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device=dev)
def pipe_time(fn, reps=10):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps + 4):
        ops.gaussian_blur(bgr, 5)
        gray, eq = ops.bgr2gray_equalize(bgr)
        if i >= 4: ev[i - 4][0].record()
        fn(gray)
        if i >= 4: ev[i - 4][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ts[len(ts) // 2] * 1e3, 1)
for _ in range(200):
    ops.laplacian_sharpen(bgr)
torch.cuda.synchronize()
for rows in [0, 0] + [int(x) for x in os.environ.get("ROWS", "4,8,12,16,24,32").split(",")] + [0]:
    os.environ["IPX_BAND3_ROWS"] = str(rows)
    print(f"rows={rows}: sobel", pipe_time(ops.sobel), "box3", pipe_time(ops.box3), flush=True)

#!/usr/bin/env python3
"""Dev A/B: the packed Gaussian as strip walker (IPX_BAND_ROWS unset) and on LDS band tiles (IPX_BAND_ROWS = rows per tile),
24 x 4K BGR and gray, in the pipeline (events around the call, other kernels before and after).  Needs the -DIPX_DEV library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device=dev)
gray = ops.bgr2gray(bgr)
def pipe_time(fn, reps=10):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps + 2):
        ops.bgr2gray_equalize(bgr); ops.sobel(gray)
        if i >= 2: ev[i - 2][0].record()
        fn()
        if i >= 2: ev[i - 2][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ts[len(ts) // 2] * 1e3, 1)
for _ in range(300):                                   # warm the chip up: the first measurements of a process read 10-20 % high
    ops.gaussian_blur(bgr, 5); ops.sobel(gray)
torch.cuda.synchronize()
os.environ["IPX_BAND_ROWS"] = "0"
ref5, ref3, refg = ops.gaussian_blur(bgr, 5).clone(), ops.gaussian_blur(bgr, 3).clone(), ops.gaussian_blur(gray, 5).clone()
for _rep in range(2): print("walker: g5 bgr", pipe_time(lambda: ops.gaussian_blur(bgr, 5)), "g3 bgr", pipe_time(lambda: ops.gaussian_blur(bgr, 3)), "g5 gray", pipe_time(lambda: ops.gaussian_blur(gray, 5)), flush=True)
for rows in [int(x) for x in os.environ.get("ROWS", "8,11,12,16,20,24,32").split(",")]:
    os.environ["IPX_BAND_ROWS"] = str(rows)
    ok = bool((ops.gaussian_blur(bgr, 5) == ref5).all()) and bool((ops.gaussian_blur(bgr, 3) == ref3).all()) and bool((ops.gaussian_blur(gray, 5) == refg).all())
    print(f"band rows={rows}: g5 bgr", pipe_time(lambda: ops.gaussian_blur(bgr, 5)), "g3 bgr", pipe_time(lambda: ops.gaussian_blur(bgr, 3)), "g5 gray", pipe_time(lambda: ops.gaussian_blur(gray, 5)), "ok", ok, flush=True)

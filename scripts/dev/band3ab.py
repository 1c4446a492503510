#!/usr/bin/env python3
"""Dev A/B: the 3 x 3 operators on BGR frames as strip walker (IPX_BAND3_ROWS=0) and on LDS band tiles, 24 x 4K, in the
pipeline (events around the call, other kernels before and after), chip warmed up first.  Needs the -DIPX_DEV library."""
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
    for i in range(reps + 4):
        ops.bgr2gray_equalize(bgr); ops.gaussian_blur(gray, 5)
        if i >= 4: ev[i - 4][0].record()
        fn()
        if i >= 4: ev[i - 4][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ts[len(ts) // 2] * 1e3, 1)
for _ in range(300):
    ops.laplacian_sharpen(bgr); ops.gaussian_blur(gray, 5)
torch.cuda.synchronize()
os.environ["IPX_BAND3_ROWS"] = "0"
fns = {"sharpen": ops.laplacian_sharpen, "box3": ops.box3, "sobel": ops.sobel}
refs = {k: f(bgr).clone() for k, f in fns.items()}
for rows in [0, 0] + [int(x) for x in os.environ.get("ROWS", "8,12,14,16,20").split(",")] + [0]:
    os.environ["IPX_BAND3_ROWS"] = str(rows)
    ok = all(bool((f(bgr) == refs[k]).all()) for k, f in fns.items())
    print(f"rows={rows}:", {k: pipe_time(lambda: f(bgr)) for k, f in fns.items()}, "ok", ok, flush=True)

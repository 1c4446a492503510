#!/usr/bin/env python3
"""Dev sweep (needs the -DIPX_DEV library: make -C image-processing_amd dev; IPX_LIB=image-processing_amd/lib_dev/libipx.so):
in-pipeline duration of the strip-walker kernels against rows per segment.  Every measured launch sits between two HIP
events and between launches of OTHER kernels, as in bench.py's step, so that launches cannot overlap at their edges."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
H, W = 2160, 3840
B = int(os.environ.get("B", "24"))
dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8).to(dev)
gray = ops.bgr2gray(bgr)
other = torch.randint(0, 256, (B, H, W, 1), dtype=torch.uint8, device=dev)
px = B * H * W


def pipe_time(fn, reps=12):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps + 2):
        ops.laplacian_sharpen(bgr)                       # an unrelated kernel before ...
        if i >= 2: ev[i - 2][0].record()
        fn()
        if i >= 2: ev[i - 2][1].record()
        ops.gaussian_blur(bgr, 3)                         # ... and after
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e-3


segs = [int(x) for x in os.environ.get("SEGS", "0,8,12,16,20,21,24,28,32,40,41,48,64").split(",")]
ident = list(range(255, -1, -1))
cases = [("lut_gray", 2, lambda: ops.apply_lut(other, ident)), ("lut_bgr", 6, lambda: ops.apply_lut(bgr, ident)),
         ("bgr2gray", 4, lambda: ops.bgr2gray(bgr)), ("hist_gray", 1, lambda: ops.hist(other)),
         ("sobel_gray", 2, lambda: ops.sobel(gray)), ("box3_gray", 2, lambda: ops.box3(gray)),
         ("sharpen_bgr", 6, lambda: ops.laplacian_sharpen(bgr)), ("gauss5_bgr", 6, lambda: ops.gaussian_blur(bgr, 5)),
         ("gauss3_bgr", 6, lambda: ops.gaussian_blur(bgr, 3)), ("gauss7_bgr", 6, lambda: ops.gaussian_blur(bgr, 7)),
         ("gauss5_gray", 2, lambda: ops.gaussian_blur(gray, 5))]
import time
t_end = time.time() + 1.5
while time.time() < t_end:
    ops.laplacian_sharpen(bgr); torch.cuda.synchronize()
for wpb in [int(x) for x in os.environ.get("WPBS", "0").split(",")]:
    os.environ["IPX_WPB"] = str(wpb)
    for name, bpp, fn in cases:
        row = []
        for seg in segs:
            os.environ["IPX_SEG_ROWS"] = str(seg)
            s = pipe_time(fn)
            row.append(f"{seg}:{s * 1e6:.1f}")
        print(f"B={B} wpb={wpb} {name:12s} " + " ".join(row), flush=True)

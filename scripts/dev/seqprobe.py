"""Dev probe: the 5x5 Gaussian's duration inside sustained loops with different neighbours."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch
from ipx import ops

dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (16, 2160, 3840, 3), dtype=torch.uint8, device=dev)
gray = ops.bgr2gray(bgr)
lut = list(range(255, -1, -1))

def loop(name, others, reps=60):
    evs = []
    for i in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = ops.gaussian_blur(bgr, 5); e1.record()
        keep = [f() for f in others]
        evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[10:])
    print(f"{t[len(t) // 2]:7.1f} us  gaussian in a loop with {name}", flush=True)

loop("nothing else", [])
loop("sobel(gray)", [lambda: ops.sobel(gray)])
loop("table pass on gray", [lambda: ops.apply_lut(gray, lut)])
loop("bgr2gray", [lambda: ops.bgr2gray(bgr)])
loop("bgr2gray_equalize", [lambda: ops.bgr2gray_equalize(bgr)])
loop("equalize_hist(gray)", [lambda: ops.equalize_hist(gray)])
loop("hist(gray) only", [lambda: ops.hist(gray)])
loop("bgr2gray_equalize + sobel", [lambda: ops.bgr2gray_equalize(bgr), lambda: ops.sobel(gray)])
loop("sharpen(bgr)", [lambda: ops.laplacian_sharpen(bgr)])

#!/usr/bin/env python3
"""Dev: does the second reader of the BGR batch hit the Infinity Cache when the Gaussian and the gray conversion take the
batch in chunks of a few frames (gauss(chunk), gray_hist(chunk), next chunk)?  24 x 4K BGR."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch
from ipx import ops

def time_op(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

dev = torch.device("cuda:0")
n, h, w = 24, 2160, 3840
bgr = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device=dev)
def whole():
    ops.gaussian_blur(bgr, 5); ops.bgr2gray_hist(bgr)
for _ in range(20): whole()
print(f"whole batch: gauss + gray_hist {time_op(whole):7.1f} us", flush=True)
for c in (1, 2, 3, 4, 6, 8, 12):
    def gf():
        for i in range(0, n, c):
            ops.gaussian_blur(bgr[i:i + c], 5); ops.bgr2gray_hist(bgr[i:i + c])
    def hf():
        for i in range(0, n, c):
            ops.bgr2gray_hist(bgr[i:i + c]); ops.gaussian_blur(bgr[i:i + c], 5)
    print(f"chunks of {c:2d}: gauss first {time_op(gf):7.1f} us, gray first {time_op(hf):7.1f} us", flush=True)

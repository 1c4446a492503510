"""Dev: per-op timing of BASELINE configs[2] on one GPU's share (128 x 1080p), 20 launches between two events."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch
from ipx import ops, luts

dev = torch.device("cuda:0")
n, h, w = 128, 1080, 1920
bgr = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8).to(dev)
gray = ops.bgr2gray(bgr)
px = n * h * w

def t(name, fn, bpp, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:22s} {us:8.1f} us  {px * bpp / us / 1e6:6.2f} TB/s algorithmic", flush=True)

t("bgr2gray_equalize", lambda: ops.bgr2gray_equalize(bgr), 6)
t("bgr2gray", lambda: ops.bgr2gray(bgr), 4)
t("lut gray", lambda: ops.apply_lut(gray, luts.negative()), 2)
t("hist gray", lambda: ops.hist(gray), 1)
t("equalize gray", lambda: ops.equalize_hist(gray), 2)
t("sharpen bgr", lambda: ops.laplacian_sharpen(bgr), 6)
t("gauss5 bgr", lambda: ops.gaussian_blur(bgr, 5), 6)
t("sobel gray", lambda: ops.sobel(gray), 2)
t("box gray", lambda: ops.box3(gray), 2)

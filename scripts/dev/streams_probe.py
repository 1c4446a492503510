"""Dev probe: the bench step's three ops on ONE stream vs the independent Gaussian on a second stream."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch
from ipx import ops

dev = torch.device("cuda:0")
B = 16
bgr = torch.randint(0, 256, (B, 2160, 3840, 3), dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def one():
    g5 = ops.gaussian_blur(bgr, 5)
    gray, eq = ops.bgr2gray_equalize(bgr)
    sb = ops.sobel(gray)
    return g5, eq, sb

def two():
    with torch.cuda.stream(s1):
        g5 = ops.gaussian_blur(bgr, 5)
    with torch.cuda.stream(s2):
        gray, eq = ops.bgr2gray_equalize(bgr)
        sb = ops.sobel(gray)
    return g5, eq, sb

for name, fn in (("one stream", one), ("two streams", two), ("one stream", one), ("two streams", two)):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        r = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"{name}: {dt * 1e3:.3f} ms/step  {B * 8.2944 / dt / 1e3:.1f} k MP/s", flush=True)

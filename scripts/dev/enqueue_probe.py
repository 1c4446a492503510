"""Dev probe: host time to ENQUEUE one bench step (ctypes calls + torch allocations) against its GPU time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch
from ipx import ops

dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (16, 2160, 3840, 3), dtype=torch.uint8, device=dev)
def step():
    g5 = ops.gaussian_blur(bgr, 5)
    gray, eq = ops.bgr2gray_equalize(bgr)
    sb = ops.sobel(gray)
for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / 50:.3f} ms/step, total {1e3 * (t2 - t0) / 50:.3f} ms/step")

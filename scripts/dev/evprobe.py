"""Dev probe: cost of HIP events inside the bench step (they keep consecutive kernels from overlapping at their edges)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch
from ipx import ops

dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (16, 2160, 3840, 3), dtype=torch.uint8, device=dev)

def run(mode, steps=40):
    def step():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        if mode >= 1: ev[0].record()
        g5 = ops.gaussian_blur(bgr, 5)
        if mode >= 1: ev[1].record()
        gray, eq = ops.bgr2gray_equalize(bgr)
        if mode >= 2: ev[2].record()
        sb = ops.sobel(gray)
        if mode >= 2: ev[3].record()
        return ev
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = [step() for _ in range(steps)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    g = sum(e[0].elapsed_time(e[1]) for e in evs) / steps * 1e3 if mode >= 1 else float("nan")
    return dt * 1e3, g

for rep in range(2):
    for mode, name in ((0, "no events"), (1, "events around the Gaussian only"), (2, "events around every op")):
        ms, g = run(mode)
        print(f"{name:34s} {ms:.4f} ms/step  {16 * 8.2944 / ms:.1f} k MP/s   gaussian {g:.1f} us", flush=True)

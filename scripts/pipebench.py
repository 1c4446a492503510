#!/usr/bin/env python3
"""Dev experiment: cfg2 step (gauss5 + gray/equalize + sobel) whole-batch vs chunked (Infinity-Cache-sized chunks),
eager vs HIP-graph replay."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
H, W = 2160, 3840
B = int(os.environ.get("B", "16"))
dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8).to(dev)

def step_whole(order="a"):
    g5 = ops.gaussian_blur(bgr, 5)
    gray, eq = ops.bgr2gray_equalize(bgr)
    sb = ops.sobel(gray)
    return g5, sb, eq

def make_chunked(chunk, order):
    def f():
        outs = []
        for i in range(0, B, chunk):
            x = bgr[i:i + chunk]
            if order == "gray_first":
                gray, eq = ops.bgr2gray_equalize(x)
                g5 = ops.gaussian_blur(x, 5)
                sb = ops.sobel(gray)
            elif order == "gauss_first":
                g5 = ops.gaussian_blur(x, 5)
                gray, eq = ops.bgr2gray_equalize(x)
                sb = ops.sobel(gray)
            else:  # sobel before the table pass
                g5 = ops.gaussian_blur(x, 5)
                gray = ops.bgr2gray(x)
                sb = ops.sobel(gray)
                eq = ops.equalize_hist(gray)
            outs.append((g5, sb, eq))
        return outs
    return f

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def graphed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    return g, keep

t_end = time.time() + 1.5
while time.time() < t_end:
    step_whole(); torch.cuda.synchronize()
mp = B * H * W / 1e6
t = timeit(step_whole); print(f"whole-batch eager           {t*1e6:8.1f} us  {mp/t:10.0f} MP/s", flush=True)
g, keep = graphed(step_whole); t = timeit(g.replay); print(f"whole-batch graph           {t*1e6:8.1f} us  {mp/t:10.0f} MP/s", flush=True)
for order in ("gauss_first", "gray_first", "unfused"):
    for chunk in (1, 2, 4, 8):
        if chunk > B: continue
        f = make_chunked(chunk, order)
        t = timeit(f); te = t
        g, keep = graphed(f); t = timeit(g.replay)
        print(f"chunk={chunk} {order:12s} eager {te*1e6:8.1f} us  graph {t*1e6:8.1f} us  {mp/t:10.0f} MP/s", flush=True)
        del g, keep

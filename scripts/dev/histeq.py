#!/usr/bin/env python3
"""Dev: cv2.equalizeHist on one 4096 x 32768 gray partition and on 24 x 4K gray: median of event-timed
calls with another kernel in between (cold caches, as in a pipeline) and the mean of a back-to-back loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
part = torch.randint(0, 256, (1, 4096, 32768, 1), dtype=torch.uint8, device=dev)
batch = torch.randint(0, 256, (24, 2160, 3840, 1), dtype=torch.uint8, device=dev)
other = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device=dev)
one = torch.randint(0, 256, (1, 2160, 3840, 1), dtype=torch.uint8, device=dev)
def cold(fn, n=15):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        ops.laplacian_sharpen(other)          # flushes the caches between measurements
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ts[len(ts) // 2] * 1e3, 1)
def loop(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n * 1e3, 1)
for name, x in (("partition 4096x32768", part), ("24x4K gray", batch), ("one 4K gray", one)):
    print(name, "equalize_hist: cold", cold(lambda: ops.equalize_hist(x)), "us, loop", loop(lambda: ops.equalize_hist(x)),
          "us; hist alone: cold", cold(lambda: ops.hist(x)), "loop", loop(lambda: ops.hist(x)), flush=True)

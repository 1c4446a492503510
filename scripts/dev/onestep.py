#!/usr/bin/env python3
"""Dev: the one-image configs[1] step (ipx.pipelines.configs1_one_image) eager, then replayed from a graph, then the forked
graph; 30 steps each separated by a marker kernel (torch fill) so that a kernel trace can be cut into steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
import ipx.pipelines as P
dev = torch.device("cuda:0")
one = torch.randint(0, 256, (2160, 3840, 3), dtype=torch.uint8, device=dev)
mark = torch.zeros(64, device=dev)
side = torch.cuda.Stream()
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        mark.fill_(1.0)
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ts[len(ts) // 2] * 1e3, 1)
print("eager", timed(lambda: P.configs1_one_image(ops, one)))
g = P.GraphedStep(lambda x: P.configs1_one_image(ops, x), one)
print("graph", timed(lambda: g()))
gf = P.GraphedStep(lambda x: P.configs1_one_image(ops, x, side), one)
print("graph forked", timed(lambda: gf()))

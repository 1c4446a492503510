#!/usr/bin/env python3
"""Dev: histogram pass alone on one 4096 x 32768 gray partition and on 24 x 4K gray.  Run once per IPX_HIST_SLOTS value
(read once per process) with IPX_LIB pointing at the -DIPX_DEV library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
part = torch.randint(0, 256, (1, 4096, 32768, 1), dtype=torch.uint8, device=dev)
batch = torch.randint(0, 256, (24, 2160, 3840, 1), dtype=torch.uint8, device=dev)
other = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        ops.laplacian_sharpen(other)          # flushes the caches between measurements
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ts[len(ts) // 2] * 1e3, 1)
print("slots", os.environ.get("IPX_HIST_SLOTS", "default"), "partition hist", t(lambda: ops.hist(part)), "us; 24x4K hist", t(lambda: ops.hist(batch)), "us")

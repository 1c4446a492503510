#!/usr/bin/env python3
"""Dev: the 5-tap Gaussian on 24 x 4K BGR alone, back to back, for the band-tile geometry the environment selects
(IPX_BAND_ROWS, IPX_BAND_LANES, IPX_BAND_ROUND; dev library)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (int(os.environ.get("B", "24")), 2160, 3840, 3), dtype=torch.uint8, device=dev)
ref = None
def run(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for _ in range(30): ops.gaussian_blur(bgr, 5)
ts = [run(lambda: ops.gaussian_blur(bgr, 5)) for _ in range(4)]
print({k: os.environ.get(k) for k in ("IPX_BAND_ROWS", "IPX_BAND_LANES", "IPX_BAND_ROUND")}, " ".join(f"{t:.1f}" for t in ts), "checksum", int(ops.gaussian_blur(bgr, 5).sum(dtype=torch.int64)) - int(bgr.sum(dtype=torch.int64)), flush=True)

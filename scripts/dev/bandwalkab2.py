#!/usr/bin/env python3
"""Dev (-DIPX_DEV library built with scripts/dev/experiments/bandwalk_kernel.patch applied): band walker vs band tiles for the 5-tap Gaussian, alone and inside the configs[1] step (no events
between the step's kernels), 24 x 4K BGR."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
from ipx import pipelines
dev = torch.device("cuda:0")

def time_op(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

n, h, w = 24, 2160, 3840
bgr = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device=dev)
step = lambda: pipelines.configs1_step(ops, bgr)
for _ in range(60): step()
for rnd in range(2):
    for S, slots in ((0, 0), (10, 2048), (10, 4096), (10, 8192), (15, 2048), (15, 4096), (15, 8192), (5, 4096), (5, 8192), (0, 0)):
        os.environ["IPX_BANDWALK"] = str(S); os.environ["IPX_BANDWALK_SLOTS"] = str(slots)
        print(f"S={S:2d} slots={slots:5d}: gauss alone {time_op(lambda: ops.gaussian_blur(bgr, 5)):7.1f} us   step {time_op(step):7.1f} us", flush=True)

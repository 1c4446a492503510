"""Dev probe: does a working set that fits the 256 MiB Infinity Cache stream faster than one that does not?
Times the table pass (read n bytes, write n bytes) and BGR->gray on batches of 4K frames, back to back."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch
from ipx import ops

dev = torch.device("cuda:0")
lut = list(range(255, -1, -1))
for frames in (1, 2, 4, 8, 16, 32):
    gray = torch.randint(0, 256, (frames, 2160, 3840, 1), dtype=torch.uint8, device=dev)
    bgr = torch.randint(0, 256, (frames, 2160, 3840, 3), dtype=torch.uint8, device=dev)
    for name, fn, nbytes in (("lut_gray", lambda: ops.apply_lut(gray, lut), 2 * gray.numel()),
                             ("bgr2gray", lambda: ops.bgr2gray(bgr), bgr.numel() + gray.numel())):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / reps
        print(f"{name:9s} frames={frames:2d} {nbytes / 1e6:7.1f} MB  {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)

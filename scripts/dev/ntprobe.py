"""Dev probe: table pass and fused gray+equalize with and without nt, by batch size (run once per IPX_NT_BYTES value)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch
from ipx import ops

dev = torch.device("cuda:0")
lut = torch.arange(255, -1, -1, dtype=torch.uint8, device=dev)
out = []
for frames in (4, 8, 12, 16, 20, 24, 32):
    gray = torch.randint(0, 256, (frames, 2160, 3840, 1), dtype=torch.uint8, device=dev)
    bgr = torch.randint(0, 256, (frames, 2160, 3840, 3), dtype=torch.uint8, device=dev)
    row = [frames]
    for fn in (lambda: ops.apply_lut_dev(gray, lut), lambda: ops.equalize_hist(gray), lambda: ops.bgr2gray_equalize(bgr), lambda: ops.bgr2gray(bgr), lambda: ops.apply_lut_dev(bgr, lut)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        row.append(round(e0.elapsed_time(e1) * 1e3 / 20 / frames, 2))
    print("frames %2d: us per frame  lut %.2f  equalize_gray %.2f  bgr2gray_equalize %.2f  bgr2gray %.2f  lut_bgr %.2f" % tuple(row), flush=True)

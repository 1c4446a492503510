#!/usr/bin/env python3
"""Dev: cv2.medianBlur k = 11..31 on 24 x 4K gray (and one frame): ms per launch, 5 back-to-back launches between two events.
With IPX_LIB -> the -DIPX_DEV library, IPX_MEDIAN_COLHIST=1 selects the sliding-column-histogram kernel for k >= 13."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
ks = [int(a) for a in sys.argv[1:]] or [11, 13, 15, 17, 21, 25, 31]
batch = torch.randint(0, 256, (24, 2160, 3840, 1), dtype=torch.uint8, device=dev)
one = batch[:1]
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for k in ks:
    print(f"k={k}: 24x4K {t(lambda: ops.median_blur(batch, k)):.3f} ms ({t(lambda: ops.median_blur(batch, k)) / 24:.3f} per frame); one frame {t(lambda: ops.median_blur(one, k)):.3f} ms", flush=True)

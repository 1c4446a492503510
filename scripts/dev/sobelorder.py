"""Dev probe: Sobel on 24 gray 4K frames right after the kernel that wrote them vs after an intervening table pass."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch
from ipx import ops

dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device=dev)
lut = torch.arange(255, -1, -1, dtype=torch.uint8, device=dev)
def run(between):
    ts = []
    for _ in range(30):
        gray = ops.bgr2gray(bgr)
        if between:
            eq = ops.apply_lut_dev(gray, lut)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sb = ops.sobel(gray); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts[5:])
    return v[len(v) // 2]
print(f"sobel right after bgr2gray: {run(False):.1f} us;  after bgr2gray + table pass: {run(True):.1f} us")

import os, sys
sys.path.insert(0, "image-processing_amd")
import torch
from ipx import ops
g = torch.randint(0, 256, (24, 2160, 3840, 1), dtype=torch.uint8, device="cuda:0")
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for w in (7, 9, 11, 13):
    print(f"lee w={w}: {t(lambda: ops.lee_filter(g, w)):9.1f} us per 24 x 4K", flush=True)

#!/usr/bin/env python3
"""Dev probe: equalizeHist paths in-pipeline (events around each call, other kernels between), with the table pass
walking forward (IPX_NO_REVERSE=1, dev library) or backward (default), and the Sobel that follows bgr2gray_equalize."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")


def med(ts):
    ts = sorted(ts); return ts[len(ts) // 2]


def pipe(fns, before, after, reps=10):
    """fns: list of callables run in order, each bracketed by events; before/after: unrelated kernels around the group"""
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(fns) + 1)] for _ in range(reps)]
    for i in range(reps + 2):
        before()
        for j, f in enumerate(fns):
            if i >= 2: ev[i - 2][j].record()
            f()
        if i >= 2: ev[i - 2][len(fns)].record()
        after()
    torch.cuda.synchronize()
    return [med([e[j].elapsed_time(e[j + 1]) for e in ev]) * 1e3 for j in range(len(fns))]


big = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device=dev)
other = torch.randint(0, 256, (24, 2160, 3840, 1), dtype=torch.uint8, device=dev)
part = torch.randint(0, 256, (4096, 32768), dtype=torch.uint8, device=dev)
gray24 = ops.bgr2gray(big)
state = {}
def before(): ops.gaussian_blur(big, 5)
def after(): ops.apply_lut(other, list(range(256)))
import time
t_end = time.time() + 1.0
while time.time() < t_end:
    before(); torch.cuda.synchronize()
for mode in ("reverse", "forward"):
    if mode == "forward": os.environ["IPX_NO_REVERSE"] = "1"
    else: os.environ.pop("IPX_NO_REVERSE", None)
    t = pipe([lambda: ops.equalize_hist(part)], before, after)
    print(f"{mode}: equalize_hist 4096x32768 partition {t[0]:.1f} us  ({2 * part.numel() / t[0] / 1e6:.2f} TB/s algorithmic)", flush=True)
    t = pipe([lambda: ops.equalize_hist(gray24)], before, after)
    print(f"{mode}: equalize_hist 24x4K gray {t[0]:.1f} us  ({2 * gray24.numel() / t[0] / 1e6:.2f} TB/s algorithmic)", flush=True)
    def geq(): state["g"], state["e"] = ops.bgr2gray_equalize(big)
    def sob(): state["s"] = ops.sobel(state["g"])
    t = pipe([geq, sob], before, after)
    print(f"{mode}: bgr2gray_equalize 24x4K {t[0]:.1f} us, sobel after it {t[1]:.1f} us", flush=True)

"""Dev probe: device-to-device copy rate of 398 MB buffers, the same pair repeatedly vs rotating over distinct pairs
(nothing left in the 256 MiB Infinity Cache), with torch's copy kernel and with the library's table pass (identity)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch
from ipx import ops

dev = torch.device("cuda:0")
n = 16 * 2160 * 3840 * 3
pairs = [(torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)) for _ in range(6)]

def rate(fn, reps=60):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2 * n / us / 1e6

print("torch copy, same pair     : %.1f us  %.2f TB/s" % rate(lambda i: pairs[0][1].copy_(pairs[0][0])))
print("torch copy, rotating pairs: %.1f us  %.2f TB/s" % rate(lambda i: pairs[i % 6][1].copy_(pairs[i % 6][0])))
img = [p[0].view(16, 2160, 3840, 3) for p in pairs]
ident = list(range(256))
print("table pass, same batch    : %.1f us  %.2f TB/s" % rate(lambda i: ops.apply_lut(img[0], ident)))
print("table pass, rotating      : %.1f us  %.2f TB/s" % rate(lambda i: ops.apply_lut(img[i % 6], ident)))
print("gaussian 5x5, same batch  : %.1f us  %.2f TB/s" % rate(lambda i: ops.gaussian_blur(img[0], 5)))
print("gaussian 5x5, rotating    : %.1f us  %.2f TB/s" % rate(lambda i: ops.gaussian_blur(img[i % 6], 5)))
print("bgr2gray, same batch      : %.1f us" % rate(lambda i: ops.bgr2gray(img[0]))[0])
print("bgr2gray, rotating        : %.1f us" % rate(lambda i: ops.bgr2gray(img[i % 6]))[0])

#!/usr/bin/env python3
"""Dev probe: why does the same box3 / Sobel launch take 63 / 70 us in the C++ harness (scripts/dev/bandbench.hip) and
80 / 81 us in the python harnesses?  Same events, same sizes; vary: preallocated output, neighbours, input provenance."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
import ipx._lib as L
dev = torch.device("cuda:0")
B, H, W = 24, 2160, 3840
lib = L.load()


def med(ts):
    ts = sorted(ts); return ts[len(ts) // 2]


def pipe(fn, before, after, reps=12):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps + 2):
        before()
        if i >= 2: ev[i - 2][0].record()
        fn()
        if i >= 2: ev[i - 2][1].record()
        after()
    torch.cuda.synchronize()
    return med([a.elapsed_time(b) for a, b in ev]) * 1e3


bgr = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device=dev)
gray_from_bgr = ops.bgr2gray(bgr)
gray_rand = torch.randint(0, 256, (B, H, W, 1), dtype=torch.uint8, device=dev)
out_fixed = torch.empty_like(gray_rand)
o1 = torch.randint(0, 256, (4 * B * H * W,), dtype=torch.uint8, device=dev)
o2 = torch.empty_like(o1)
st = torch.cuda.current_stream().cuda_stream


def raw(name, src, dst):
    L.call(name, st, src.data_ptr(), dst.data_ptr(), B, H, W, 1, W, W * H, W, W * H, None, None)


nb = {"copy": (lambda: o2.copy_(o1), lambda: o2.copy_(o1)),
      "lut_bgr": (lambda: ops.apply_lut(bgr, list(range(256))), lambda: ops.apply_lut(bgr, list(range(256)))),
      "sharpen": (lambda: ops.laplacian_sharpen(bgr), lambda: ops.gaussian_blur(bgr, 3))}
t_end = time.time() + 1.0
while time.time() < t_end:
    ops.laplacian_sharpen(bgr); torch.cuda.synchronize()
for nbn, (bf, af) in nb.items():
    for srcn, src in (("gray_rand", gray_rand), ("gray_from_bgr", gray_from_bgr)):
        t1 = pipe(lambda: raw("ipx_box3_u8", src, out_fixed), bf, af)
        t2 = pipe(lambda: ops.box3(src), bf, af)
        t3 = pipe(lambda: raw("ipx_sobel_u8", src, out_fixed), bf, af)
        t4 = pipe(lambda: ops.sobel(src), bf, af)
        print(f"neighbours={nbn:8s} src={srcn:14s} box3 raw/fixed-out {t1:6.1f}  box3 ops {t2:6.1f}  sobel raw {t3:6.1f}  sobel ops {t4:6.1f}", flush=True)
print("ptrs", hex(gray_rand.data_ptr()), hex(gray_from_bgr.data_ptr()), hex(out_fixed.data_ptr()))

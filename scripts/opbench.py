#!/usr/bin/env python3
"""Per-operator timing sweep on one GPU (development tool; HIP events on torch's current stream).
usage: python scripts/opbench.py [--batch 16] [--seg 0,24,32,48,64,96,128]"""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops

H, W = 2160, 3840

def time_op(fn, reps=40):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seg", default="0,16,24,32,48,64,96,128")
    ap.add_argument("--what", default="conv,eq,point")
    ap.add_argument("--wpb", default="0")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.batch
    bgr = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8).to(dev)
    gray = ops.bgr2gray(bgr)
    px = B * H * W
    t_end = __import__("time").time() + 1.5          # clock ramp: keep the GPU busy before timing anything
    while __import__("time").time() < t_end:
        ops.laplacian_sharpen(bgr); torch.cuda.synchronize()
    def rep(name, bpp, fn, extra=""):
        s = time_op(fn)
        print(f"{name:28s} {extra:16s} {s*1e6:9.1f} us  {px*bpp/s/1e9:8.1f} GB/s  {px*bpp/s/8e12*100:5.1f}% of 8TB/s", flush=True)
    what = a.what.split(",")
    if "conv" in what:
        for seg, wpb in [(int(x), int(w)) for x in a.seg.split(",") for w in a.wpb.split(",")]:
            os.environ["IPX_SEG_ROWS"] = str(seg)
            os.environ["IPX_WPB"] = str(wpb)
            e = f"seg={seg} wpb={wpb}"
            rep("sharpen_bgr", 6, lambda: ops.laplacian_sharpen(bgr), e)
            rep("sobel_gray", 2, lambda: ops.sobel(gray), e)
            rep("box3_gray", 2, lambda: ops.box3(gray), e)
            rep("gauss3_bgr", 6, lambda: ops.gaussian_blur(bgr, 3), e)
            rep("gauss5_bgr", 6, lambda: ops.gaussian_blur(bgr, 5), e)
            rep("gauss5_gray", 2, lambda: ops.gaussian_blur(gray, 5), e)
            rep("gauss7_bgr", 6, lambda: ops.gaussian_blur(bgr, 7), e)
        os.environ["IPX_SEG_ROWS"] = "0"
    if "big" in what:
        b8 = torch.randint(0, 256, (2, 4320, 7680, 3), dtype=torch.uint8).to(dev)
        p8 = 2 * 4320 * 7680
        for k in (9, 15, 31):
            s = time_op(lambda: ops.gaussian_blur(b8, k), reps=3)
            print(f"gauss{k}_8k_bgr {s*1e6:9.1f} us {p8*6/s/1e9:8.1f} GB/s", flush=True)
    if "lee" in what:
        for wnd in (3, 5, 9, 11, 15, 31):
            g1 = gray[:1]
            s = time_op(lambda: ops.lee_filter(g1, wnd), reps=5)
            print(f"lee w={wnd:2d} one 4K gray frame {s*1e6:9.1f} us  {H*W/s/1e6:9.1f} MP/s", flush=True)
    if "median" in what:
        g1 = gray[:1]
        for k in (3, 5, 9, 15, 31):
            s = time_op(lambda: ops.median_blur(g1, k), reps=3)
            print(f"median k={k:2d} one 4K gray frame {s*1e6:9.1f} us  {H*W/s/1e6:9.1f} MP/s", flush=True)
        se = ops.ellipse_se(5)
        s = time_op(lambda: ops.morph_close(ops.morph_open(g1, se), se), reps=3)
        print(f"open+close 5x5 ellipse one 4K gray frame {s*1e6:9.1f} us", flush=True)
        s = time_op(lambda: ops.denoise_stats(g1[0, :, :, 0], "lee"), reps=3)
        print(f"denoise_stats one 4K gray frame (blocking, 2 host round trips) {s*1e6:9.1f} us", flush=True)
    if "eq" in what:
        for nb in (1, 2, 4, 8, 16):
            if nb > B: break
            g = gray[:nb]
            s = time_op(lambda: ops.equalize_hist(g))
            s2 = time_op(lambda: ops.hist(g))
            p = nb * H * W
            print(f"equalize n={nb:3d} {s*1e6:8.1f} us ({s*1e6/nb:6.1f}/img) {p*2/s/1e9:8.1f} GB/s | hist {s2*1e6:8.1f} us {p/s2/1e9:8.1f} GB/s(read)", flush=True)
    if "eq" in what:
        rep("bgr2gray_equalize (6 B/px)", 6, lambda: ops.bgr2gray_equalize(bgr))
    if "point" in what:
        rep("bgr2gray", 4, lambda: ops.bgr2gray(bgr))
        rep("lut_gray", 2, lambda: ops.apply_lut(gray, list(range(255, -1, -1))))
        rep("lut_bgr", 6, lambda: ops.apply_lut(bgr, list(range(255, -1, -1))))
        rep("bgr2gray_lut", 4, lambda: ops.bgr2gray_lut(bgr, list(range(255, -1, -1))))
        rep("torch_copy_bgr", 6, lambda: bgr.clone())

if __name__ == "__main__":
    main()

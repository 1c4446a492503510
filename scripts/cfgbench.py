#!/usr/bin/env python3
"""BASELINE.json configs[2..4] on ONE GPU (development record, not bench.py lines; HIP events on torch's stream).
  cfg3: batch of 1080p RGB frames through the full pipeline (gray, negative, threshold, gamma, log, piecewise, hist-eq,
        colour Laplacian sharpen, 5x5 Gaussian); --frames per GPU (1024 / 8 = 128 is one GPU's share of the node's batch)
  cfg4: one 8K RGB frame, separable Gaussian 15x15 and 31x31
  cfg5: 32768 x 32768 gray image: hist-eq + one 3x3 conv on the whole image (what 8 row partitions do together), plus
        one 4096-row partition with halo rows and a separately reduced histogram (what ONE of 8 GPUs does)
usage: python scripts/cfgbench.py [--what 3,4,5] [--frames 128]"""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import numpy as np
import torch
import ipx.ops as ops
import ipx.luts as luts


def timed(fn, reps=5, warm=2):
    out = None
    for _ in range(warm):
        out = None                              # (drop the previous outputs first: two live sets would make the allocator grow
        out = fn()                              # inside the timed region)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = None
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="3,4,5")
    ap.add_argument("--frames", type=int, default=128)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    what = a.what.split(",")
    res = {}
    if "3" in what:
        n, h, w = a.frames, 1080, 1920
        bgr = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8).to(dev)
        import ipx.pipelines as P
        def pipe():                                                  # the pipeline the config test checks against the oracle
            return P.full_pipeline(ops, bgr)                         # (per-image log tables; five table modes in one pass)
        s = timed(pipe, reps=10, warm=3)
        mp = n * h * w / 1e6
        algb = n * h * w * (4 + 7 + 6 + 6)                          # per operator: gray + histogram 4, six tables on one read 1 + 6, sharpen 6,
                                                                     # gauss 6 (the pipeline itself moves 21: the Gaussian and the gray batch share a read)
        res["cfg3"] = {"frames": n, "ms": round(s * 1e3, 3), "MP_per_s": round(mp / s, 1), "alg_GBps": round(algb / s / 1e9, 1)}
        print("cfg3", res["cfg3"], flush=True)
        del bgr
    if "4" in what:
        img = torch.randint(0, 256, (1, 4320, 7680, 3), dtype=torch.uint8).to(dev)
        for k in (15, 31):
            s = timed(lambda: ops.gaussian_blur(img, k), reps=10)
            res[f"cfg4_k{k}"] = {"us": round(s * 1e6, 1), "MP_per_s": round(4320 * 7680 / 1e6 / s, 1), "alg_GBps": round(4320 * 7680 * 6 / s / 1e9, 1)}
            print(f"cfg4 k={k}", res[f"cfg4_k{k}"], flush=True)
        del img
    if "5" in what:
        H = W = 32768
        g = torch.Generator(device="cpu").manual_seed(5)
        img = torch.empty((H, W), dtype=torch.uint8, device=dev)
        for y0 in range(0, H, 4096):                                # fill in 128 MiB slabs (host memory stays small)
            img[y0:y0 + 4096] = torch.randint(0, 256, (4096, W), dtype=torch.uint8, generator=g).to(dev)
        s_eq = timed(lambda: ops.equalize_hist(img), reps=3, warm=1)
        s_cv = timed(lambda: ops.box3(img), reps=3, warm=1)
        eq, hist, lut = ops.equalize_hist(img, return_aux=True)
        assert int(hist.sum()) == H * W
        # one of 8 row partitions: rows 4096..8191 with one halo row each side; its histogram; table from the GLOBAL histogram
        y0, y1 = 4096, 8192
        part = img[y0:y1]
        s_part_cv = timed(lambda: ops.box3(part, top_halo=img[y0 - 1:y0], bottom_halo=img[y1:y1 + 1]), reps=5)
        s_part_h = timed(lambda: ops.hist(part), reps=5)
        s_part_lut = timed(lambda: ops.apply_lut_dev(part, lut[0]), reps=5)
        whole = ops.box3(img)
        ok_cv = torch.equal(ops.box3(part, top_halo=img[y0 - 1:y0], bottom_halo=img[y1:y1 + 1]), whole[y0:y1])
        hsum = sum(ops.hist(img[i * 4096:(i + 1) * 4096]).to(torch.int64) for i in range(8))
        ok_h = torch.equal(hsum, hist[0].to(torch.int64))
        ok_eq = torch.equal(ops.apply_lut_dev(part, ops.equalize_lut(hsum.to(torch.int32))), eq[y0:y1])
        mp = H * W / 1e6
        res["cfg5"] = {"whole_equalize_ms": round(s_eq * 1e3, 3), "whole_box3_ms": round(s_cv * 1e3, 3),
                       "whole_MP_per_s_eq": round(mp / s_eq, 1), "whole_MP_per_s_conv": round(mp / s_cv, 1),
                       "partition_box3_us": round(s_part_cv * 1e6, 1), "partition_hist_us": round(s_part_h * 1e6, 1),
                       "partition_lut_us": round(s_part_lut * 1e6, 1),
                       "partition_conv_equals_whole": bool(ok_cv), "partition_hists_sum_to_global": bool(ok_h),
                       "partition_equalize_equals_whole": bool(ok_eq)}
        print("cfg5", res["cfg5"], flush=True)
    if "pcie" in what:
        # configs[1] with the boundary handing over HOST buffers: pinned H2D of 16 BGR frames, the 3-op step, pinned D2H of
        # the three outputs (Gaussian BGR, equalized gray, Sobel gray), double-buffered on two streams
        B, h, w = 16, 2160, 3840
        hin = [torch.randint(0, 256, (B, h, w, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
        hout = [(torch.empty((B, h, w, 3), dtype=torch.uint8).pin_memory(), torch.empty((B, h, w, 1), dtype=torch.uint8).pin_memory(),
                 torch.empty((B, h, w, 1), dtype=torch.uint8).pin_memory()) for _ in range(2)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        def run(iters):
            for it in range(iters):
                k = it & 1
                with torch.cuda.stream(streams[k]):
                    d = hin[k].to(dev, non_blocking=True)
                    g5 = ops.gaussian_blur(d, 5)
                    gray, eq = ops.bgr2gray_equalize(d)
                    sb = ops.sobel(gray)
                    hout[k][0].copy_(g5, non_blocking=True); hout[k][1].copy_(eq, non_blocking=True); hout[k][2].copy_(sb, non_blocking=True)
            torch.cuda.synchronize()
        run(2)
        import time
        t0 = time.perf_counter(); run(8); dt = (time.perf_counter() - t0) / 8
        res["cfg2_host_buffers"] = {"ms_per_16_frames": round(dt * 1e3, 2), "MP_per_s": round(B * h * w / 1e6 / dt, 1),
                                    "host_GBps": round((B * h * w * (3 + 3 + 1 + 1)) / dt / 1e9, 1)}
        print("pcie", res["cfg2_host_buffers"], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

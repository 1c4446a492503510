"""Dev tool (GPU box): randomized parity sweep of the HIP library against the CPU oracle.

Random shapes (incl. widths around the 16-byte / 1 KiB strip / 128-byte tile edges), channel counts, batch sizes and
ROW-STRIDED views (a window of a wider buffer), every operator.  Exact equality everywhere.  The fixed-shape cases live
in tests/test_gpu_parity.py; this looks for the shapes nobody thought of.
usage: python tests/tools/fuzz_parity.py [--seconds 90] [--seed 0] [--ops lee,median]"""
import argparse
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")  # repo root (tests/tools/ -> ../..)
for p in ("image-processing_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch

import coracle as CO
import oracle as O
from ipx import ops

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=90)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--ops", default="gray,lut,eq,grayeq,sharpen,box3,sobel,gauss,gauss,lee,median,morph,halo,stats,png,gray16,gaussgray,sobellut,gsg")
ap.add_argument("--big", action="store_true", help="images of up to ~3 MP and 3000 rows (slower oracle, fewer cases)")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda:0")

EDGE_W = [1, 2, 5, 15, 16, 17, 31, 33, 63, 64, 65, 127, 128, 129, 341, 342, 343, 1023, 1024, 1025, 1365, 2047, 2049,
          1376, 2064, 2752, 4096, 4112, 1440]            # rows of whole 16-byte lanes >= 4 KiB at 1-4 channels: the LDS band-tile kernels


def rand_shape(max_px=400_000):
    if a.big:
        w = int(rng.choice(EDGE_W[8:])) if rng.random() < 0.3 else int(rng.integers(100, 4200))
        h = int(rng.integers(1, max(2, min(3000, 3_000_000 // w))))
        return h, w
    w = int(rng.choice(EDGE_W)) if rng.random() < 0.5 else int(rng.integers(1, 2200))
    h = int(rng.integers(1, max(2, min(300, max_px // w))))
    return h, w


def make(n, h, w, c, strided):
    """device tensor (n, h, w, c) -- optionally a window of a wider / taller buffer -- and its numpy value"""
    kind = rng.integers(0, 4)
    if kind == 0:
        img = rng.integers(0, 256, (n, h, w, c), dtype=np.uint8)
    elif kind == 1:
        img = (rng.integers(0, 2, (n, h, w, c)) * 255).astype(np.uint8)
    elif kind == 2:
        img = np.clip(rng.normal(128, 12, (n, h, w, c)), 0, 255).astype(np.uint8)
    else:
        img = np.full((n, h, w, c), int(rng.integers(0, 256)), np.uint8)
    if not strided:
        return torch.from_numpy(img).to(dev), img
    px, py = int(rng.integers(1, 40)), int(rng.integers(0, 3))
    buf = torch.from_numpy(rng.integers(0, 256, (n, h + py, w + px, c), dtype=np.uint8)).to(dev)
    x0 = int(rng.integers(0, px + 1))
    view = buf[:, :h, x0:x0 + w, :]
    view.copy_(torch.from_numpy(img).to(dev))
    return view, img


def per_image(fn, img):
    return np.stack([fn(img[i] if img.shape[-1] > 1 else img[i, :, :, 0]).reshape(img.shape[1], img.shape[2], -1)
                     for i in range(img.shape[0])])


def check(name, got, want, info):
    got = got.cpu().numpy().reshape(want.shape)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        print(f"MISMATCH {name} {info}: {len(bad)} values, first at {bad[0].tolist()} got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}", flush=True)
        return 1
    return 0


t_end = time.time() + a.seconds
cases = fails = 0
counts = {}
while time.time() < t_end:
    op = str(rng.choice(a.ops.split(",")))
    n = int(rng.integers(1, 4))
    strided = bool(rng.random() < 0.5)
    h, w = rand_shape()
    if op == "gray":
        c = int(rng.choice([3, 4]))
        t, img = make(n, h, w, c, strided)
        fails += check(op, ops.bgr2gray(t), per_image(O.bgr2gray_u8, img), (n, h, w, c, strided))
    elif op == "lut":
        c = int(rng.choice([1, 3]))
        t, img = make(n, h, w, c, strided)
        lut = rng.permutation(256).astype(np.uint8)
        fails += check(op, ops.apply_lut(t, lut), lut[img], (n, h, w, c, strided))
    elif op == "eq":
        t, img = make(n, h, w, 1, strided)
        fails += check(op, ops.equalize_hist(t), per_image(O.equalize_hist_u8, img), (n, h, w, strided))
    elif op == "grayeq":
        c = int(rng.choice([3, 4]))
        t, img = make(n, h, w, c, strided)
        gray, eq = ops.bgr2gray_equalize(t)
        g = per_image(O.bgr2gray_u8, img)
        fails += check(op + ".gray", gray, g, (n, h, w, c, strided))
        fails += check(op + ".eq", eq, per_image(O.equalize_hist_u8, g), (n, h, w, c, strided))
    elif op in ("sharpen", "box3", "sobel"):
        c = int(rng.choice([1, 2, 3, 4]))
        t, img = make(n, h, w, c, strided)
        fn = {"sharpen": (ops.laplacian_sharpen, O.laplacian_sharpen_u8), "box3": (ops.box3, O.box3_u8), "sobel": (ops.sobel, O.sobel_u8)}[op]
        fails += check(op, fn[0](t), per_image(fn[1], img), (n, h, w, c, strided))
    elif op == "gauss":
        c = int(rng.choice([1, 3, 4]))
        k = int(rng.choice([1, 3, 5, 7, 9, 11, 13, 15, 17, 21, 25, 31]))
        t, img = make(n, h, w, c, strided)
        fails += check(f"gauss{k}", ops.gaussian_blur(t, k), per_image(lambda x: O.gaussian_blur_u8(x, k), img), (n, h, w, c, strided))
    elif op == "gaussw":                                     # 9..31 taps on inputs of 2 MB and more: the wave-per-strip kernel
        c = int(rng.choice([1, 3, 4]))                       # (interior segments + frame pieces; C oracle: the sizes are large)
        k = int(rng.choice([9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31]))
        w = int(rng.choice([683, 700, 1024, 1365, 1366, 2048, 2731, 3000])) if rng.random() < 0.5 else int(rng.integers(300, 3200))
        h = int(rng.integers(max(40, (2 << 20) // (n * w * c) + 1), max(41, (9 << 20) // (n * w * c))))
        t, img = make(n, h, w, c, strided)
        fails += check(f"gaussw{k}", ops.gaussian_blur(t, k), per_image(lambda x: CO.gaussian_blur_u8(x, k), img), (n, h, w, c, strided))
    elif op == "eqbig":                                      # images cut into 65+ histogram blocks: the in-launch finish (tickets)
        w = int(rng.integers(900, 4200))
        h = int(rng.integers((1 << 21) // w + 1, (6 << 20) // w))
        n = int(rng.integers(1, 3))
        t, img = make(n, h, w, 1, strided)
        fails += check(op, ops.equalize_hist(t), per_image(CO.equalize_hist_u8, img), (n, h, w, strided))
        c = int(rng.choice([3, 4]))
        t, img = make(1, h // 2 + 1, w, c, strided)
        gray, eq = ops.bgr2gray_equalize(t)
        g = per_image(CO.bgr2gray_u8, img)
        fails += check(op + ".gray", gray, g, (1, h // 2 + 1, w, c, strided))
        fails += check(op + ".grayeq", eq, per_image(CO.equalize_hist_u8, g), (1, h // 2 + 1, w, c, strided))
    elif op == "png":                                        # PNG reconstruction: random filter types per row
        bpp = int(rng.choice([1, 2, 3, 4, 6, 8]))
        w = max(1, min(w, 3000) // bpp)
        _, img = make(1, h, w * bpp, 1, False)
        img = img.reshape(h, w * bpp)
        plan = rng.integers(0, 5, h) if rng.random() < 0.6 else np.full(h, int(rng.integers(0, 5)))
        raw = CO.png_filter(img, bpp, plan.astype(np.uint8))
        got, status = ops.png_unfilter(torch.from_numpy(np.ascontiguousarray(raw).reshape(-1)).to(dev), h, w, bpp)
        fails += check(f"png{bpp}", got, img, (h, w, bpp))
        if int(status.item()) != 0:
            print(f"MISMATCH png status {int(status.item())} {(h, w, bpp)}", flush=True)
            fails += 1
    elif op == "gray16":                                     # 16-bit samples -> gray -> >> 8
        c = int(rng.choice([1, 2, 3, 4]))
        be, rgb = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
        w = max(1, min(w, 2100))
        raw = rng.integers(0, 256, (n, h, w * c * 2), dtype=np.uint8)
        if rng.random() < 0.2:
            raw[:] = 255
        t = torch.from_numpy(raw).to(dev)
        if strided:
            buf = torch.zeros((n, h, w * c * 2 + 21), dtype=torch.uint8, device=dev)
            buf[:, :, :w * c * 2] = t
            t = buf[:, :, :w * c * 2]
        fails += check(f"gray16.{c}", ops.gray_from_u16(t, w, c, big_endian=be, rgb_order=rgb), O.gray_u8_from_u16_bytes(raw, w, c, be, rgb), (n, h, w, c, be, rgb, strided))
    elif op == "gaussgray":                                  # Gaussian + gray on one read of a BGR batch (band-tile shapes and fall-backs)
        k = int(rng.choice([3, 5, 5, 5, 7]))
        w = int(rng.choice([1366, 1376, 1392, 2048, 2064, 2752, 3840, 4112])) if rng.random() < 0.7 else max(1, min(w, 1500))
        h = int(rng.integers(1, 40))
        t, img = make(n, h, w, 3, strided)
        g5, gray = ops.gaussian_blur_gray(t, k)
        fails += check(f"gaussgray{k}.blur", g5, per_image(lambda x: CO.gaussian_blur_u8(x, k), img), (n, h, w, strided))
        fails += check(f"gaussgray{k}.gray", gray, per_image(CO.bgr2gray_u8, img), (n, h, w, strided))
    elif op == "gsg":                                        # Gaussian + Laplacian sharpening + gray on one read of a BGR batch
        k = int(rng.choice([5, 5, 5, 3]))
        w = int(rng.choice([1184, 1376, 1920, 2048, 2752, 3840])) if rng.random() < 0.7 else max(1, min(w, 1500))
        h = int(rng.integers(1, 40))
        t, img = make(n, h, w, 3, strided)
        g5, sh, gray = ops.gaussian_sharpen_gray(t, k)
        fails += check(f"gsg{k}.blur", g5, per_image(lambda x: CO.gaussian_blur_u8(x, k), img), (n, h, w, strided))
        fails += check(f"gsg{k}.sharpen", sh, per_image(CO.laplacian_sharpen_u8, img), (n, h, w, strided))
        fails += check(f"gsg{k}.gray", gray, per_image(CO.bgr2gray_u8, img), (n, h, w, strided))
    elif op == "sobellut":                                   # Sobel + per-image table pass on one read of a gray batch
        t, img = make(n, h, w, 1, strided)
        tabs = np.stack([rng.permutation(256).astype(np.uint8) for _ in range(n)])
        sb, tb = ops.sobel_lut(t, torch.from_numpy(tabs).to(dev))
        fails += check(op + ".sobel", sb, per_image(O.sobel_u8, img), (n, h, w, strided))
        fails += check(op + ".table", tb, np.stack([tabs[i][img[i]] for i in range(n)]), (n, h, w, strided))
        hst, lut = ops.hist_tables(t)
        want = np.stack([O.equalize_lut_from_hist(np.bincount(img[i].ravel(), minlength=256)) for i in range(n)])
        fails += check(op + ".tables", lut, want, (n, h, w, strided))
    elif op == "gaussbig":                                   # the generic two-pass path (k > 31)
        c = int(rng.choice([1, 1, 2, 3, 4]))
        k = int(rng.choice([33, 35, 45, 63, 101, 127, 255]))
        h, w = min(h, 200), min(w, 600)
        t, img = make(n, h, w, c, strided)
        fails += check(f"gauss{k}", ops.gaussian_blur(t, k), per_image(lambda x: O.gaussian_blur_u8(x, k), img), (n, h, w, c, strided))
    elif op == "lee":
        h, w = min(h, 120), min(w, 300)
        win = int(rng.choice([1, 2, 3, 3, 4, 5, 5, 7, 9, 9, 10, 11, 11, 15, 31]))
        t, img = make(n, h, w, 1, strided)
        fails += check(f"lee{win}", ops.lee_filter(t, win), per_image(lambda x: O.lee_filter_u8(x, win), img), (n, h, w, strided))
    elif op == "median":
        h, w = min(h, 150), min(w, 700)
        k = int(rng.choice([1, 3, 3, 5, 5, 5, 7, 7, 7, 9, 9, 9, 11, 11, 11, 13, 13, 15, 15, 17, 19, 21, 23, 25, 27, 29, 31]))
        t, img = make(n, h, w, 1, strided)
        fails += check(f"median{k}", ops.median_blur(t, k), per_image(lambda x: O.median_blur_u8(x, k), img), (n, h, w, strided))
    elif op == "halo":
        # a row block [y0, y1) of a taller image with r halo rows from its neighbours == the whole-image result cropped
        c = int(rng.choice([1, 3]))
        which = str(rng.choice(["sharpen", "box3", "sobel", "gauss"]))
        k = int(rng.choice([3, 5, 7, 9, 15, 31, 33, 47])) if which == "gauss" else 3
        r = k // 2
        h = max(h, 3 * r + 3)
        t, img = make(1, h, w, c, False)
        y0 = 0 if rng.random() < 0.3 else int(rng.integers(r, h - r - 1))
        lo = y0 + 1                                          # blocks shorter than r are legal next to an image border:
        if lo >= h - r + 1 and rng.random() < 0.7:           # the reflection then reaches into the other side's halo
            y1 = h
        elif lo >= h - r + 1:
            continue
        else:
            y1 = h if rng.random() < 0.3 else int(rng.integers(lo, h - r + 1))
        if (y0 > 0 and y1 < h and y1 - y0 < r) or y1 <= y0:    # between two neighbours a block must hold r rows (RowPartition)
            continue
        local = t[0, y0:y1]
        top = t[0, y0 - r:y0] if y0 > 0 else None
        bot = t[0, y1:y1 + r] if y1 < h else None
        if which == "gauss":
            got = ops.gaussian_blur(local, k, top_halo=top, bottom_halo=bot)
            want = per_image(lambda x: O.gaussian_blur_u8(x, k), img)[0, y0:y1]
        else:
            fn = {"sharpen": (ops.laplacian_sharpen, O.laplacian_sharpen_u8), "box3": (ops.box3, O.box3_u8), "sobel": (ops.sobel, O.sobel_u8)}[which]
            got = fn[0](local, top_halo=top, bottom_halo=bot)
            want = per_image(fn[1], img)[0, y0:y1]
        fails += check(f"halo.{which}{k}", got, want, (h, w, c, y0, y1))
        if which != "gauss" and c == 1:                      # the same block with a table pass on the same read (ipx_conv3_lut_u8)
            name = {"sharpen": "laplacian_sharpen", "box3": "box3", "sobel": "sobel"}[which]
            tab = rng.permutation(256).astype(np.uint8)
            f2, tb = ops.conv3_lut(name, local, torch.from_numpy(tab).to(dev), top_halo=top, bottom_halo=bot)
            fails += check(f"halo.{which}+table.filter", f2, want, (h, w, c, y0, y1))
            fails += check(f"halo.{which}+table.table", tb, tab[img[0, y0:y1]], (h, w, c, y0, y1))
    elif op == "stats":
        t, img = make(1, h, w, 1, strided)
        got = ops.denoise_stats(t[0, :, :, 0] if not strided else t[0, :, :, 0].contiguous())
        want = O.denoise_stats(img[0, :, :, 0])
        if any(got[k2] != want[k2] for k2 in want):
            print(f"MISMATCH stats {(h, w)}: {got} vs {want}", flush=True)
            fails += 1
    else:
        h, w = min(h, 150), min(w, 700)
        m = int(rng.choice([1, 3, 5, 7, 9, 15, 31]))
        se = ops.ellipse_se(m)
        t, img = make(n, h, w, 1, strided)
        fails += check(f"open{m}", ops.morph_open(t, se), per_image(lambda x: O.morph_open_u8(x, se), img), (n, h, w, strided))
        fails += check(f"close{m}", ops.morph_close(t, se), per_image(lambda x: O.morph_close_u8(x, se), img), (n, h, w, strided))
    cases += 1
    counts[op] = counts.get(op, 0) + 1
    if cases % 50 == 0:
        print(f"{cases} cases, {fails} mismatches", flush=True)
print(f"done: {cases} cases {counts}, {fails} mismatches")
sys.exit(1 if fails else 0)

#!/usr/bin/env python3
"""Dev: the PNG reconstruction kernel on one 4K frame (RGB, gray, RGBA): GPU time per launch (20 launches in one graph would
reuse the workspace; here 10 event-timed launches), for photograph-like content filtered by Pillow's encoder and for rows
forced to Paeth."""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from PIL import Image
import ipx.ops as ops, ipx.codec as codec
import coracle as CO
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
h, w = 2160, 3840
yy, xx = np.mgrid[0:h, 0:w]
def smooth(c):
    base = 128 + 60 * np.sin(xx / 97.0) * np.cos(yy / 131.0)
    img = np.stack([base + 20 * np.sin((xx + 50 * k) / 41.0) for k in range(c)], -1) + rng.normal(0, 4, (h, w, c))
    return np.clip(img, 0, 255).astype(np.uint8)
def t(raw, bpp, n=10):
    r = torch.from_numpy(np.frombuffer(raw, np.uint8).copy()).to(dev)
    ops.png_unfilter(r, h, w, bpp); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); out, st = ops.png_unfilter(r, h, w, bpp); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ts[len(ts) // 2], 3), int(st.item()), out
for c in (3, 1, 4):
    img = smooth(c)
    buf = io.BytesIO(); Image.fromarray(img[:, :, 0] if c == 1 else img).save(buf, format="PNG", compress_level=1)
    t0 = time.perf_counter(); got = codec.png_scanlines(buf.getvalue()); t_inf = time.perf_counter() - t0
    t0 = time.perf_counter(); ref = np.array(Image.open(io.BytesIO(buf.getvalue()))); t_pil = time.perf_counter() - t0
    ms, st, out = t(got[4], c)
    ok = bool((out.cpu().numpy().reshape(ref.shape) == ref).all())
    types = np.bincount(np.frombuffer(got[4], np.uint8)[::1 + w * c], minlength=5).tolist()
    print(f"{c} ch: file {len(buf.getvalue()) / 1e6:.1f} MB, inflate {t_inf * 1e3:.1f} ms, Pillow decode {t_pil * 1e3:.1f} ms, GPU un-filter {ms} ms, status {st}, equal {ok}, filter types {types}", flush=True)
    raw = CO.png_filter(img.reshape(h, -1), c, np.full(h, 4, np.uint8))
    print(f"   all rows Paeth: GPU un-filter {t(raw, c)[0]} ms", flush=True)

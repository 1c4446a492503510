"""Dev tool (THIS container only: imports the real reference from /root/reference through oracle/make_golden.py's
codec stand-in).  Looks for images where the vectorised oracle restatement of the Lee filter differs from the
reference's own per-pixel loop (server.py:711-732), and from a scalar emulation of the HIP kernel's operation order."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")  # repo root (tests/tools/ -> ../..)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import make_golden as MG
import oracle as O

sys.dont_write_bytecode = True
MG._install_codec_standin()
sys.path.insert(0, MG.REF)
cwd = os.getcwd()
os.chdir(MG.REF)
try:
    import server
finally:
    os.chdir(cwd)

f32 = np.float32


def kernel_order(img, window):
    """ipx_lee.hip's arithmetic, scalar by scalar"""
    H, W = img.shape
    k = window // 2
    out = img.copy()
    n = (2 * k + 1) ** 2
    for y in range(k, H - k):
        for x in range(k, W - k):
            win = img[y - k:y + k + 1, x - k:x + k + 1].astype(np.float32).ravel()
            mean = f32(f32(win.sum(dtype=np.float64)) / f32(n))
            d = (win - mean).astype(np.float32)
            var = f32(O.pairwise_sum_f32((d * d).astype(np.float32)) / f32(n))
            if var > 0:
                q = f32(f32(mean * mean) / f32(mean + f32(1e-12)))
                kl = f32(var / f32(var + q))
                o = f32(mean + f32(kl * f32(f32(img[y, x]) - mean)))
            else:
                o = mean
            out[y, x] = np.uint8(min(max(o, f32(0)), f32(255)))
    return out


seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
t0 = time.time()
for it in range(40):
    window = int(rng.choice([3, 5, 7]))
    kind = it % 3
    if kind == 0:
        img = (rng.integers(0, 2, (100, 300)) * 255).astype(np.uint8)
    elif kind == 1:
        img = np.clip(rng.normal(128, 12, (100, 300)), 0, 255).astype(np.uint8)
    else:
        img = rng.integers(0, 256, (100, 300), dtype=np.uint8)
    ref = server.lee_filter(img, window_size=window)
    orc = O.lee_filter_u8(img, window)
    ker = kernel_order(img, window)
    a, b = int((ref != orc).sum()), int((ref != ker).sum())
    print(f"it {it} kind {kind} window {window}: reference vs oracle {a} diffs, reference vs kernel-order {b} diffs ({time.time() - t0:.0f}s)", flush=True)
    if a or b:
        for name, other in (("oracle", orc), ("kernel-order", ker)):
            for (y, x) in np.argwhere(ref != other)[:3]:
                print(f"   {name} at ({y},{x}): reference {ref[y, x]} vs {other[y, x]}")

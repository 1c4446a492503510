"""dev: where does the wave kernel differ from the oracle on a small image?  usage: wavedbg.py h w cn k"""
import os, sys
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(root, "image-processing_amd")); sys.path.insert(0, root)
import torch, ipx.ops as ops
from oracle import oracle as O
h, w, cn, k = [int(v) for v in sys.argv[1:5]]
rng = np.random.default_rng(10)
img = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
want = O.gaussian_blur_u8(img, k)
for rep in range(3):
    got = ops.gaussian_blur(torch.from_numpy(img).cuda(), k).cpu().numpy()
    d = (got != want).reshape(h, -1)
    ys, xs = np.nonzero(d)
    print("rep", rep, "mismatches", int(d.sum()), "rows", sorted(set(ys.tolist()))[:40], "byte cols", (int(xs.min()), int(xs.max())) if len(xs) else None, flush=True)

"""Dev: per-wave start/end stamps of one 5x5 Gaussian launch (IPX_CONV_TRACE) -> occupancy over time."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import numpy as np
import torch
from ipx import ops

dev = torch.device("cuda:0")
bgr = torch.randint(0, 256, (16, 2160, 3840, 3), dtype=torch.uint8, device=dev)
for _ in range(5):
    ops.gaussian_blur(bgr, 5)
torch.cuda.synchronize()
path = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "wavetrace.bin")
os.environ["IPX_CONV_TRACE"] = path
ops.gaussian_blur(bgr, 5)
del os.environ["IPX_CONV_TRACE"]
raw = np.fromfile(path, dtype=np.uint64)
nmain = int(raw[0])
t_all = raw[2:].reshape(-1, 2).astype(np.int64)
tb = t_all[nmain:]
tb = tb[(tb[:, 0] > 0) & (tb[:, 1] > 0)]
t = t_all[:nmain]
t = t[(t[:, 0] > 0) & (t[:, 1] > 0)]
t0 = min(t[:, 0].min(), tb[:, 0].min() if len(tb) else t[:, 0].min())
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0          # us (100 MHz)
if len(tb):
    bs, be = (tb[:, 0] - t0) / 100.0, (tb[:, 1] - t0) / 100.0
    print(f"border waves: {len(tb)}; first start {bs.min():.1f} us, last end {be.max():.1f} us, lifetime median {np.median(be - bs):.2f} us (p90 {np.percentile(be - bs, 90):.2f})")
print(f"{len(t)} waves; kernel span {en.max():.1f} us; wave lifetime median {np.median(en - st):.1f} us (p10 {np.percentile(en - st, 10):.1f}, p90 {np.percentile(en - st, 90):.1f})")
print(f"starts: 50% by {np.percentile(st, 50):.1f} us, first-round (5120th wave) by {np.sort(st)[min(5119, len(st) - 1)]:.1f} us")
edges = np.arange(0, en.max() + 5, 5.0)
for a in edges[:-1]:
    active = int(((st < a + 5) & (en > a)).sum())
    started = int(((st >= a) & (st < a + 5)).sum())
    ended = int(((en >= a) & (en < a + 5)).sum())
    print(f"  {a:5.0f}-{a + 5:3.0f} us: active {active:5d}  started {started:5d}  ended {ended:5d}")

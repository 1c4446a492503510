"""Dev probe: why is the 5x5 Gaussian slower inside the bench step than back to back?  Times it (HIP events per call)
after different predecessors on the same stream."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch
from ipx import ops

dev = torch.device("cuda:0")
B = 16
bgr = torch.randint(0, 256, (B, 2160, 3840, 3), dtype=torch.uint8, device=dev)
bgr2 = torch.randint(0, 256, (B, 2160, 3840, 3), dtype=torch.uint8, device=dev)
gray = ops.bgr2gray(bgr)
lut = list(range(255, -1, -1))

def timed(pre, src, reps=30):
    ts = []
    for _ in range(reps):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gaussian_blur(src, 5)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

cases = [("nothing (sync before)", lambda: None, bgr),
         ("same gaussian", lambda: ops.gaussian_blur(bgr, 5), bgr),
         ("gaussian of ANOTHER batch", lambda: ops.gaussian_blur(bgr2, 5), bgr),
         ("sobel gray (133 MB in, 133 out)", lambda: ops.sobel(gray), bgr),
         ("table pass on gray", lambda: ops.apply_lut(gray, lut), bgr),
         ("bgr2gray of the same batch", lambda: ops.bgr2gray(bgr), bgr),
         ("bgr2gray of ANOTHER batch", lambda: ops.bgr2gray(bgr2), bgr)]
for name, pre, src in cases:
    print(f"{timed(pre, src):7.1f} us  after {name}", flush=True)

# sustained load: does the per-launch time drift (clock / power management)?
print("sustained back-to-back Gaussian, 100 launches per sample:")
torch.cuda.synchronize()
for chunk in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        ops.gaussian_blur(bgr, 5)
    e1.record()
    torch.cuda.synchronize()
    print(f"  chunk {chunk}: {e0.elapsed_time(e1) * 10:.1f} us per launch", flush=True)
print("sustained bench step (gauss, gray+equalize, sobel), per-op events, 300 steps:")
evs = []
for i in range(300):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(); ops.gaussian_blur(bgr, 5); ev[1].record(); g, e = ops.bgr2gray_equalize(bgr); ev[2].record(); ops.sobel(g); ev[3].record()
    evs.append(ev)
torch.cuda.synchronize()
for lo in (0, 20, 100, 200, 280):
    s = [sum(evs[i][j].elapsed_time(evs[i][j + 1]) for i in range(lo, lo + 20)) / 20 * 1e3 for j in range(3)]
    print(f"  steps {lo}-{lo + 19}: gauss {s[0]:.1f}  gray+eq {s[1]:.1f}  sobel {s[2]:.1f} us", flush=True)
